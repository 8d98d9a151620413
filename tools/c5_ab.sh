#!/bin/bash
# A/B of BASELINE configs[4] under environment switches: bash tools/c5_ab.sh "VAR=a" "VAR=b" ...   (GPU box, via gpurun)
for v in "$@"; do env $v python bench.py --config c5 --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d.get('value'),1), 'steps/s', d.get('ms_per_step'))"; done
