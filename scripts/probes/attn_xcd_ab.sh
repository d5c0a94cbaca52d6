for m in 0 2 1; do echo "== YAT_SDPA_XCD=$m"; YAT_HIP_LIB=yat_amd/build/variants/libyat_tune.so YAT_SDPA_XCD=$m timeout -k 10 200 python scripts/attn_bench.py 2>&1 | grep sdpa; done
