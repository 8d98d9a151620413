#!/bin/bash
# A/B of the headline step under environment switches: bash tools/headline_ab.sh "VAR=a" "VAR=b" ...   (GPU box, via gpurun)
for v in "$@"; do env $v python bench.py --steps 300 --no-cpu-baseline --no-native-leg --no-dropin --no-secondary --no-traffic --no-dry-leg --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'steps/s', round(d['ms_per_step']*1e3,1), 'us')"; done
