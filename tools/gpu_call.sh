#!/bin/bash
# Run one recorded GPU call of tools/gpu_calls.manifest by id (r04c17, r05c1, ...): tools/gpu_call.sh <id> [args...]
# (the 58 tools/r04_callNN.sh one-shot scripts of round 4, folded into one manifest; `tools/gpu_call.sh --list` shows the ids)
M="$(dirname "$0")/gpu_calls.manifest"
if [ "$1" = "--list" ] || [ -z "$1" ]; then grep "^#== " "$M" | sed 's/^#== //' | tr '\n' ' '; echo; exit 0; fi
ID="$1"; shift
BODY=$(awk -v id="#== $ID" '$0==id{p=1;next} /^#== /{p=0} p' "$M")
[ -z "$BODY" ] && { echo "no call '$ID' in $M" >&2; exit 2; }
bash -c "$BODY" gpu_call "$@"
