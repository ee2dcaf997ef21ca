L=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
for v in new old; do
  if [ $v = old ]; then export TE_ATTN_FWD=old; else unset TE_ATTN_FWD; fi
  rm -f gpurun_out/parity_report.jsonl
  TE_RELPROP_LIB=$L timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -x -k "config1" 2>&1 | tail -2
  python - <<EOF
import json
for l in open('gpurun_out/parity_report.jsonl'):
    r=json.loads(l)
    if 'e2e_vs_fp64' in r.get('name',''):
        print("$v", {k:(round(v,5) if isinstance(v,float) else v) for k,v in r.items() if k not in ('per_sample','name')})
        for s in r['per_sample']: print('   ', {k:(round(v,5) if isinstance(v,float) else v) for k,v in s.items()})
EOF
done
