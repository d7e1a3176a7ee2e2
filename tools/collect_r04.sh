#!/bin/bash
# Round-4 evidence in one GPU call: bash tools/collect_r04.sh <git-head>   -> gpurun_out/r04/...
HEAD=${1:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04; mkdir -p $O
bash $R/tools/collect_profiles.sh r04 $HEAD > $O/collect_profiles.log 2>&1; echo "headline done"
cd /tmp && export TMPDIR=/tmp
for wl in hm_config3 module_config5 real_tiles reinhard_f32; do
  timeout -k 10 300 python3 $R/bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl done"
done
bash $R/tools/collect_siblings.sh r04/sib > $O/collect_siblings.log 2>&1; echo "sibling traces done"
# PMC traffic of the other configurations: (name, driver, algorithmic bytes per step)
while read name drv alg; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${name}_f -o f -- python3 $R/tools/$(echo $drv) > $O/pmc_${name}_f.log 2>&1 &&
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${name}_w -o w -- python3 $R/tools/$(echo $drv) > $O/pmc_${name}_w.log 2>&1 &&
  python3 $R/tools/pmc_traffic_any.py $O/pmc_${name}_f $O/pmc_${name}_w 1 $alg > $O/traffic_$name.json; echo "pmc $name done"
done <<LIST
config5_bf16 prof_config5.py\ bf16 154140672
u8_64x512 prof_config5.py\ u8 100663296
hm_config3 prof_hm.py 402653184
reinhard_f32 prof_reinhard.py\ f32 402653184
LIST
ls $O
