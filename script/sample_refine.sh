#!/bin/bash
# R stage (MF-MDM refine) on MI355X: stands where the reference's script/sample_refine.sh:33-38 stands.
#
#   script/sample_refine.sh [-y] [-n] <split> <refine checkpoint> <model_name> [extra launcher flags ...]
#
# Same three positional arguments and launcher argument list as the reference wrapper (the split's process range and segment cache,
# the refine checkpoint, the <split>/<model_name> offset under which script/sample.sh left the G samples, --commit).
# -y skips the confirmation prompt, -n prints the command and exits (dry run).
set -u
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
yes=0; dry=0
while [ $# -gt 0 ]; do
    case "$1" in
        -y|--yes) yes=1; shift ;;
        -n|--dry-run) dry=1; shift ;;
        -h|--help) sed -n '2,8p' "${BASH_SOURCE[0]}" | sed 's/^# \{0,1\}//'; exit 0 ;;
        *) break ;;
    esac
done
if [ $# -lt 3 ]; then
    echo "usage: script/sample_refine.sh [-y] [-n] <split> <refine checkpoint> <model_name> [extra flags]" >&2
    exit 2
fi
split="$1"; weight="$2"; name="$3"; shift 3
printf 'split:      %s\nmodel:      %s\nmodel_name: %s\n' "$split" "$weight" "$name"

cmd=(python -m oakink2_tamf_amd.launch.sample_refine
     --data.process_range "?(file:./asset/split/$split.txt)"
     --data.cache_dict_filepath "common/save_cache_dict/main/cache/$split.pkl"
     --debug.model_weight_filepath "$weight"
     --debug.sample_save_offset "$split/$name"
     --commit "$@")

if [ "$dry" = 1 ]; then
    printf '%q ' "${cmd[@]}"; echo
    exit 0
fi
if [ "$yes" != 1 ]; then
    read -r -p "Refine the samples of split '$split' under $split/$name with '$weight'? [y/N] " answer
    case "$(printf '%s' "$answer" | tr 'A-Z' 'a-z')" in
        y|yes) ;;
        *) echo "aborted"; exit 1 ;;
    esac
fi
export PYTHONPATH="$here/oakink2-tamf_amd${PYTHONPATH:+:$PYTHONPATH}"
exec "${cmd[@]}"
