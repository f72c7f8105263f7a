#!/bin/bash
# Everything profiles/ holds for a round, from the final binary, in one call on the GPU box:
#   tools/collect_round_profiles.sh r04      -> gpurun_out/r04_* (copy the summaries into profiles/)
# kernel trace (steady table, sequence, stats) of the headline bench and of the HRNet-w48 workload,
# MFMA-busy per kernel class, SQ / TA counters of the encoder sampler, its HBM traffic (PMC passes
# of their own: no trace domains mixed with --pmc), the per-shape GEMM census, the sampler alone.
P=$1
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_bench.sh ${P}_bench_T7B4_bf16x3 && echo "trace ok"
SEQ_MIN_US=40 bash tools/profile_bench.sh ${P}_hrnet_w48 --backbone hrnet_w48 && echo "hrnet trace ok"
bash tools/pmc_bench_mfma.sh ${P}_mfma > /dev/null && cp gpurun_out/${P}_mfma/mfma_util.txt gpurun_out/${P}_bench_T7B4_mfma_util.txt && echo "mfma ok"
bash tools/pmc_enc_tile.sh ${P}_pmc_enc > gpurun_out/${P}_pmc_enc_tile.txt 2>&1 && echo "enc pmc ok"
bash tools/pmc_bench_enc.sh ${P}_enc_traffic > gpurun_out/${P}_enc_traffic.log 2>&1 && echo "enc traffic ok"
python3 tools/gemm_census.py 3 > gpurun_out/${P}_gemm_shapes.txt 2>&1 && echo "census ok"
SEQ_MIN_US=0 bash tools/profile_bench.sh ${P}_t3b1 --frames 3 --clips 1 --steps 6 && echo "configs[1] trace ok"
SEQ_MIN_US=0 bash tools/profile_bench.sh ${P}_t3b1_750x1333 --frames 3 --clips 1 --steps 6 --height 750 --width 1333 && echo "PoseTrack canvas trace ok"
SEQ_MIN_US=40 bash tools/profile_bench.sh ${P}_swin_l_t3 --backbone swin_l --frames 3 --clips 1 && echo "swin trace ok"
python3 tools/gemm_census.py 3 0 3 1 r50 > gpurun_out/${P}_gemm_shapes_t3b1.txt 2>&1 && echo "census t3 ok"
python3 tools/gemm_census.py 3 0 15 1 r50 fp16 > gpurun_out/${P}_gemm_shapes_t15_fp16.txt 2>&1 && echo "census t15 fp16 ok"
python3 tools/gemm_census.py 3 0 3 1 swin_l > gpurun_out/${P}_gemm_shapes_swin_l_t3.txt 2>&1 && echo "census swin ok"
python3 tools/bench_kernels.py --frames 28 --sigma 0.9 --enc-only --prepared --ablate > gpurun_out/${P}_enc_kernels_28frames.txt 2>&1 && echo "enc kernels ok"
rm -rf gpurun_out/${P}_mfma/p1 gpurun_out/${P}_pmc_enc/p1 gpurun_out/${P}_pmc_enc/p2 gpurun_out/${P}_pmc_enc/p3 gpurun_out/${P}_enc_traffic/p1 gpurun_out/${P}_enc_traffic/p2
ls -la gpurun_out | grep ${P}_ | head -30
