# usage (GPU box): bash tools/ab_multi.sh "ENV_1" "ENV_2" ...  -- step time under each environment, two rounds interleaved
F="--no-cpu-baseline --no-fk --no-kernel-timing --no-phase1 --steps 100 --warmup 20"
run() { env $1 python bench.py $F 2> gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],4), d.get('loss_total'))"; }
for i in 1 2; do for e in "$@"; do run "$e"; done; done
