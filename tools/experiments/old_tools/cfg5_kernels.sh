cd "$(dirname "$0")/.."
R=$PWD; cd /tmp; export TMPDIR=/tmp
for it in 4096; do
rm -rf /tmp/c5; ( cd $R && ISO_RASTER_ITEMS=$it timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/c5 -- python tools/configs_bench.py > /tmp/c5.log 2>&1 )
python $R/tools/rocprof_summary.py $(find /tmp/c5 -name "*.db" | head -1) /tmp/c5.txt > /dev/null
echo "== items $it"; grep -E "k_raster|k_splat_backward|k_z_|k_bin|k_splat_front|k_brick_h|k_grad" /tmp/c5.txt | cut -c1-60,87-140
grep cfg5_splat /tmp/c5.log | cut -c1-120
done
