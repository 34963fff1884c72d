cd "$(dirname "$0")/.."
for v in ry1 ry2; do echo "== $v"; ISO_DEV_LIB=tools/variants/libiso_$v.so tools/heavy_time.sh | head -1; ISO_DEV_LIB=tools/variants/libiso_$v.so tools/seq_cmd.sh hv_$v >/dev/null 2>&1; grep k_splat_backward_heavy gpurun_out/hv_${v}_kernel_stats.txt | cut -c1-50,87-140; done
