# in-situ walk of some knobs only, starting from the shipped choices (GPU box).  TUNE_RT=72 runs on the image's ROCm 7.2 runtime
# (graphs destroyed normally); default: the runtime bench.py runs on (replaced graphs are parked: keep the walk short)
set -e
cd $GRAFT_REPO_ROOT
export HMVAE_TUNE_DEFAULTS=1
cp hm_vae_amd/assets/tune_mi355x.json gpurun_out/tune_new.json
if [ "${TUNE_RT:-}" = "72" ]; then export LD_PRELOAD=/opt/rocm/lib/libamdhip64.so.7; fi
timeout -k 10 ${TUNE_TIMEOUT:-1000} python3 tools/tune_in_situ.py --only ${TUNE_ONLY:-w} --passes ${TUNE_PASSES:-1} --out gpurun_out/tune_new.json ${TUNE_ARGS:-} > gpurun_out/tune_w.log 2>&1
