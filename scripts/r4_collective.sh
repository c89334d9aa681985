#!/bin/bash
# the multi-GPU code path on ONE GPU: torchrun with one rank, RCCL process group, the gradient all-reduce forced (world 1), both graph forms
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
for extra in "--force-collective" "--force-collective --split-graph" "--force-collective --overlap" "--force-collective --full-graph"; do
  out=$(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline $extra 2>/tmp/err.log | tail -1)
  python -c "
import json,sys
try:
    d=json.loads('''$out''')
    print('$extra ->', d['ms_per_step'], d.get('parity',{}).get('pass'), d.get('config',{}).get('parallelism'), d.get('graph'))
except Exception as e:
    print('$extra -> FAILED', e); print(open('/tmp/err.log').read()[-800:])
"
done
