#!/bin/bash
# A/B on ONE box: alternate the default library and a variant (tools/ab_build.sh <name>) ROUNDS times.
# usage (on the GPU box, repo root): tools/ab_run.sh <name> [rounds] -- <python command printing one JSON line with "value">
name=$1; rounds=${2:-3}; shift 2; [ "$1" == "--" ] && shift
for i in $(seq $rounds); do
  for v in base $name; do
    if [ $v == base ]; then unset QUINN_AMD_LIB; else export QUINN_AMD_LIB=$PWD/quinn_amd/lib/libquinn_amd_$name.so; fi
    out=$("$@" 2>/dev/null | tail -1)
    echo "$v $(echo "$out" | python3 -c 'import sys,json
d=json.loads(sys.stdin.read())
if "value" in d:
    r=d.get("roofline",{})
    print(round(d["value"]), r.get("kernel_ms"), d.get("extras",{}).get("grad_evals_per_s"))
else:
    print(json.dumps(d))')"
  done
done
