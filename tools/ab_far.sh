for X in 40 34 30 26; do
  MPMC_THOLE_FAR_X=$X python bench.py --cpu-baseline none --no-extra-passes --steps 10 --warmup 3 > gpurun_out/b_far$X.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/b_far$X.json')); print('X=$X', round(d['value'],1), d['obs_rd_es_pol_vdw'][2], d['roofline']['tile_pairs'], d['kernel_ms'])"
done
MPMC_THOLE_FAR_X=40 python tools/kernel_ab.py "x40:MPMC_ONE_STREAM=1,MPMC_THOLE_FAR_X=40" "x32:MPMC_ONE_STREAM=1,MPMC_THOLE_FAR_X=32" "x28:MPMC_ONE_STREAM=1,MPMC_THOLE_FAR_X=28"
