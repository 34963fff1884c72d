#!/bin/bash
# headline + cfg3a figures of bench.py for environment variants: tools/bench_ab.sh "VAR=val" "VAR=val" ...   ("-" = none)
cd "$(dirname "$0")/.."
for e in "$@"; do
  if [ "$e" == "-" ]; then ENVV=""; else ENVV="$e"; fi
  env $ENVV python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - "$e" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print("%-28s headline %.3f ms (%.2f Mp/s, frac %.4f)  cfg3a %.4f ms  opapi %.2f ms" % (sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["cfg3a_analytic_sdf"]["ms_per_step"], d["operator_api"]["ms_per_step"]))
PY
done
