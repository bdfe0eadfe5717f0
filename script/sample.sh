#!/bin/bash
# G stage (MF-MDM generate) on MI355X: stands where the reference's script/sample.sh:33-41 stands.
#
#   script/sample.sh [-y] [-n] <split> <checkpoint> <model_name> [extra launcher flags ...]
#
# Same three positional arguments and the same launcher argument list as the reference wrapper (obj_embedding.yml + arch_mdm_l.yml,
# the split's process range and segment cache, the checkpoint, the <split>/<model_name> output offset, --commit); the module is this
# repo's HIP launcher.  -y skips the confirmation prompt (the reference always asks), -n prints the command and exits (dry run).
# Devices: the reference pins --runtime.device_id 0,1,2,3; here the launcher's default is one worker per visible GPU, and
# DEVICE_ID=0,1,2,3 script/sample.sh ... restores the pin.
set -u
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
yes=0; dry=0
while [ $# -gt 0 ]; do
    case "$1" in
        -y|--yes) yes=1; shift ;;
        -n|--dry-run) dry=1; shift ;;
        -h|--help) sed -n '2,10p' "${BASH_SOURCE[0]}" | sed 's/^# \{0,1\}//'; exit 0 ;;
        *) break ;;
    esac
done
if [ $# -lt 3 ]; then
    echo "usage: script/sample.sh [-y] [-n] <split> <checkpoint> <model_name> [extra flags]" >&2
    exit 2
fi
split="$1"; weight="$2"; name="$3"; shift 3
printf 'split:      %s\nmodel:      %s\nmodel_name: %s\n' "$split" "$weight" "$name"

cmd=(python -m oakink2_tamf_amd.launch.sample
     --cfg "$here/config/obj_embedding.yml"
     --data.process_range "?(file:./asset/split/$split.txt)"
     --data.cache_dict_filepath "common/save_cache_dict/main/cache/$split.pkl"
     --cfg "$here/config/arch_mdm_l.yml"
     --debug.model_weight_filepath "$weight"
     --debug.sample_save_offset "$split/$name")
if [ -n "${DEVICE_ID:-}" ]; then cmd+=(--runtime.device_id "$DEVICE_ID"); fi
cmd+=(--commit "$@")

if [ "$dry" = 1 ]; then
    printf '%q ' "${cmd[@]}"; echo
    exit 0
fi
if [ "$yes" != 1 ]; then
    read -r -p "Sample split '$split' with '$weight' into common/sample/.../$split/$name? [y/N] " answer
    case "$(printf '%s' "$answer" | tr 'A-Z' 'a-z')" in
        y|yes) ;;
        *) echo "aborted"; exit 1 ;;
    esac
fi
export PYTHONPATH="$here/oakink2-tamf_amd${PYTHONPATH:+:$PYTHONPATH}"
exec "${cmd[@]}"
