#!/bin/bash
# quick bench line: tools/bq.sh <bench args>  -> value, per-stage ms per step, ms_per_step, roofline frac, trace-identity error
python bench.py --cpu-sample 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value'],1), d.get('stage_ms_per_step'), d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('max_trace_identity_err_4096_rows'))
"
