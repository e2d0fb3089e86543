for kb in 96 120 144; do echo "LDS_KB=$kb"; for i in 1 2; do AVSI_COOP_FINE_LDS_KB=$kb python bench.py --mode train --batch 32 --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"; done; done
AVSI_COOP_FINE_LDS_KB=120 python bench.py --batch 32 --steps 50 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | cut -c100-250
