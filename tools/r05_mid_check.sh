set -e
mkdir -p gpurun_out
L=gpurun_out/r05_mid_check.log
: > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 600 python tests/skip1_check.py >> $L 2>&1; }
run ZK_PIPE_MAX_PAIRS=1 ZK_CHECK_SIZES=3,4,7,10,12,14,16,18
run ZK_PIPE_MAX_PAIRS=1 ZK_PIPE_MID_MAX_PAIRS=131072 ZK_LEAD_MIN_PAIRS=1 ZK_SKIP1_MIN_PAIRS=1 ZK_QUAD_MAX_PAIRS=0 ZK_CHECK_SIZES=11,13,15,17,19
run ZK_CHECK_SIZES=14,16,18,20
tail -4 $L
timeout -k 10 900 python tools/ab_prover.py ZK_PIPE_MID_MAX_PAIRS=0 ZK_PIPE_MID_MAX_PAIRS=8192 ZK_PIPE_MID_MAX_PAIRS=16384 ZK_PIPE_MID_MAX_PAIRS=32768 ZK_PIPE_MID_MAX_PAIRS=65536 ZK_PIPE_MID_MAX_PAIRS=131072 ZK_PIPE_MID_MAX_PAIRS=262144 > gpurun_out/r05_mid_ab.log 2>&1
cat gpurun_out/r05_mid_ab.log
