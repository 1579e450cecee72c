# The 16-bit hand-over of the qkv-row gradient in the single-pass training legs (csrc/vmm_common.h VMM_DQKV16) against a library built with fp32 rows:
#   python tools/build_ab.py temporal_block_bwd linattn_block_bwd qkv_bwd -DVMM_DQKV16=0 && gpurun -- 'bash tools/ab_dqkv16.sh'
# kernel tests of the three variants, per-parameter gradient digests of both libraries (must agree bit for bit), step times alternating.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB=$PWD/videometamaterials_amd/libvmm_hip_ab.so
(timeout 900 python -m pytest tests/test_gpu_block_bwd.py tests/test_gpu_kernels.py -q -x -k "fused_temporal_block_backward or fused_linear_attention_block_backward or fused_to_qkv_backward" 2>&1 | tail -5) 
for p in fp16 bf16; do
VMM_DIGEST_OUT=/tmp/dig_$p.json python tools/check_dqkv16.py $p 2>&1 | grep -E "^$p"
VMM_DIGEST_CMP=/tmp/dig_$p.json VMM_DQKV16=0 VMM_LIB_PATH=$AB python tools/check_dqkv16.py $p 2>&1 | grep -E "^$p"
done
for i in $(seq ${AB_TIMING:-2}); do
python tools/time_train_modes.py fp16 bf16 2>&1 | grep -E "^(fp16|bf16)" | cut -c1-400
VMM_DQKV16=0 VMM_LIB_PATH=$AB python tools/time_train_modes.py fp16 2>&1 | grep -E "^(fp16|bf16)" | sed 's/^/AB(fp32 dqkv) /' | cut -c1-400
done
