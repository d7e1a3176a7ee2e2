#!/bin/bash
# Kernel traces of the sibling paths (HM config 3, Reinhard f32, Macenko config 5 and uint8) -> gpurun_out/<tag>/siblings_kernel_stats.txt
TAG=${1:-r03_sib}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/siblings_kernel_stats.txt
for what in "prof_hm.py" "prof_reinhard.py f32" "prof_config5.py bf16" "prof_config5.py u8"; do
  name=$(echo $what | tr ' .' '__')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o kt -- python3 $R/tools/$what > $O/$name.log 2>&1 || exit 1
  echo "== tools/$what" >> $O/siblings_kernel_stats.txt
  python3 $R/tools/profile_summary.py $O/$name 20 >> $O/siblings_kernel_stats.txt
done
cat $O/siblings_kernel_stats.txt
