timeout 900 python -m pytest tests/test_splat_gpu.py tests/test_round4_gpu.py tests/test_round3_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3
VARIANTS="head default" tools/ko_rank_raster.sh siren; VARIANTS="head default" tools/ko_rank_raster.sh sphere
tools/ab_cycle.sh "k_raster" head default head default
for v in head default; do if [ $v == default ]; then unset ISO_DEV_LIB; else export ISO_DEV_LIB=tools/variants/libiso_$v.so; fi; tools/seq_cmd.sh s_$v siren >/dev/null 2>&1; echo "$v siren: $(grep -E "k_raster" gpurun_out/s_${v}_sequence.txt | awk '{print $3}' | tr '\n' ' ')"; done
