"""Build container: the order of the memory / matrix / synchronisation instructions of one kernel in a device-ISA dump
(run-length compressed) - what tools/isa_stats.py counts, in sequence.   python tools/isa_order.py /tmp/tamf.s '<regex>' [max items]"""
import re
import sys


def body(path, pat):
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and re.search(pat, m.group(1)):
            j = i
            while not lines[j].startswith(".Lfunc_end"):
                j += 1
            return m.group(1), lines[i:j]
    raise SystemExit("no kernel matches " + pat)


def main():
    name, b = body(sys.argv[1], sys.argv[2])
    maxn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    keep = ("v_mfma", "ds_read", "ds_write", "global_load", "global_store", "global_atomic", "s_waitcnt", "s_barrier", "s_setprio", "buffer_")
    ops = []
    for l in b:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        if t[0].startswith(keep):
            ops.append(t[0] + (" " + " ".join(t[1:]) if t[0] == "s_waitcnt" else ""))
    out, prev, cnt = [], None, 0
    for o in ops + [None]:
        if o == prev:
            cnt += 1
            continue
        if prev:
            out.append(f"{prev} x{cnt}" if cnt > 1 else prev)
        prev, cnt = o, 1
    print(name[:140])
    print(" | ".join(out[:maxn]))


if __name__ == "__main__":
    main()
