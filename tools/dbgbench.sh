for d in 0 32 64 96 4; do echo "DBG=$d"; VLMC_LORA_DBG=$d timeout -k 10 100 python tools/bench_lora_gemm.py --only v7b.qkvo --iters 24 2>&1 | grep v7b | cut -d'|' -f2,3,5; done
