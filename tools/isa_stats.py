"""Build container: instruction mix and register / scratch use of one kernel in a device-ISA dump.
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -S --cuda-device-only -o /tmp/tamf.s oakink2-tamf_amd/csrc/tamf_hip.hip
    python tools/isa_stats.py /tmp/tamf.s '<regex on the mangled name>' [mnemonic ...]"""
import collections
import re
import sys


def kernel_bodies(path, pat):
    lines = open(path).read().split("\n")
    rx = re.compile(pat)
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m and rx.search(m.group(1)):
            j = i
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            k = j
            meta = {}
            while k < len(lines) and k < j + 200:
                mm = re.match(r"\s*; (NumVgprs|NumAgprs|ScratchSize|Occupancy|SGPRSpill|VGPRSpill|codeLenInByte|LDSByteSize)\S*:? (\d+)", lines[k])
                if mm:
                    meta[mm.group(1)] = int(mm.group(2))
                k += 1
            yield m.group(1), lines[i:j], meta
            i = j
        i += 1


def main():
    path, pat = sys.argv[1:3]
    want = sys.argv[3:] or ["v_mfma", "ds_read_b128", "global_load_lds", "global_store", "global_atomic", "scratch_", "s_waitcnt", "s_barrier",
                            "v_cvt_pk", "v_max3_f32", "v_exp_f32", "v_rcp_f32"]
    for name, body, meta in kernel_bodies(path, pat):
        c = collections.Counter()
        for l in body:
            t = l.strip().split()
            if t and not t[0].startswith((".", ";")) and not t[0].endswith(":"):
                c[t[0]] += 1
        print(name[:150])
        print("   instructions", sum(c.values()), meta)
        print("   " + "  ".join(f"{w}:{sum(v for k, v in c.items() if k.startswith(w))}" for w in want))


if __name__ == "__main__":
    main()
