# rocprofv3 kernel traces of the segmentation bench configurations (which kernels fill the step besides the forward)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${1:-r04zb}; mkdir -p $OUT
for cfg in semantic hovernet; do
  rm -rf /tmp/rp_$cfg
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$cfg -- \
      python $R/bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_trace_$cfg.json 2> /dev/null)
  python $R/scripts/prof_summarize.py /tmp/rp_$cfg $OUT/${TAG}_trace_${cfg}_summary.txt > /dev/null
  head -30 $OUT/${TAG}_trace_${cfg}_summary.txt | cut -c1-170
  cut -c1-300 $OUT/${TAG}_trace_$cfg.json
done
