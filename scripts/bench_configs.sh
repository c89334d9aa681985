O=gpurun_out/r02_configs; mkdir -p $O
for a in "--config cfg2" "--config cfg2 --channels 128" "--config cfg3" "--config cfg4" "--config cfg4 --dtype fp8" "--config cfg4 --batch 32 --dtype fp8" "--variant strided" "--dtype bf16" "--dtype fp32"; do
  n=$(echo $a | tr -d ' -' )
  timeout 300 python bench.py $a --no-cpu-baseline --no-kernel-timer --no-eager > $O/$n.json 2> $O/$n.err
  python - "$a" $O/$n.json <<'PY'
import sys, json
l=[x for x in open(sys.argv[2]).read().strip().split('\n') if x.startswith('{')]
if not l: print(sys.argv[1], 'NO OUTPUT'); sys.exit()
d=json.loads(l[-1]); p=d.get('parity',{})
print('%-44s ms %8.3f  seq/s %9.1f  parity %s %s'%(sys.argv[1], d['ms_per_step'], d['value'], p.get('vs_fp32_hip',{}).get('max_abs'), p.get('pass')))
PY
done
