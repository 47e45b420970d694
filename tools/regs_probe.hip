// Register / scratch census of the kernels on the config-2 launch list without building the whole library (seconds instead of minutes):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/regs_probe.s tools/regs_probe.hip -Rpass-analysis=kernel-resource-usage   (= python tools/regs_probe.py)
//   python tools/isa_blocks.py /tmp/regs_probe.s <kernel>     (where the scratch_load / scratch_store sit)
#include "../emojivoice_amd/csrc/ev_kernels.h"
template __global__ void ln_mlp_h16_kernel<0>(const MlpParams);
template __global__ void ln_qkv_h16_kernel<0>(const MlpParams);
template __global__ void conv_h16_bal_kernel<128, 128, 2, 2, 1, 9>(const ConvParams);
template __global__ void conv_h16_bal_kernel<128, 128, 2, 2, 3, 9>(const ConvParams);
template __global__ void conv_h16_bal_kernel<128, 128, 2, 2, 1, 12>(const ConvParams);
template __global__ void conv_h16_bal_kernel<128, 128, 2, 2, 3, 12>(const ConvParams);
template __global__ void conv_h16_kernel<128, 128, 2, 2, 1>(const ConvParams);
template __global__ void conv_h16_kernel<128, 128, 2, 2, 3>(const ConvParams);
template __global__ void conv_h16_kernel<64, 128, 2, 2, 1>(const ConvParams);
template __global__ void resblock_pair_h16_kernel<2, 2, 1>(const PairParams);
template __global__ void resblock_pair_h16_kernel<1, 4, 1>(const PairParams);
template __global__ void resblock_pair_h16_kernel<4, 1, 1>(const PairParams);
template __global__ void resblock_pair_h16_kernel<2, 2, 3>(const PairParams);
template __global__ void resblock_pair_h16_kernel<1, 4, 3>(const PairParams);
// builds of the other arithmetic settings / batch-1 paths that spilled in round 3
template __global__ void ln_mlp_kernel<0, 3>(const MlpParams);
template __global__ void ln_mlp_kernel<0, 2>(const MlpParams);
template __global__ void ln_mlp_split_kernel<6>(const MlpParams);
template __global__ void conv_split_bal_kernel<128, 128, 2, 2, 1, 6>(const ConvParams);
template __global__ void conv_split_bal_kernel<128, 128, 2, 2, 3, 6>(const ConvParams);
template __global__ void conv_gemm_bal_kernel<64, 192, 2, 2, 1>(const ConvParams);
template __global__ void conv_gemm_bal_kernel<64, 192, 2, 2, 3>(const ConvParams);
template __global__ void conv_gemm_sk_kernel<4, false, 1, 4>(const ConvParams);
template __global__ void conv_gemm_sk_kernel<4, false, 2, 4>(const ConvParams);
template __global__ void conv_gemm_sk_kernel<4, false, 3, 4>(const ConvParams);
template __global__ void conv_gemm_sk_kernel<4, false, 0, 4>(const ConvParams);
template __global__ void conv_gemm_sk_kernel<4, true, 0, 4>(const ConvParams);
template __global__ void conv_gemm_kernel<128, 192, 2, 2, false, true, 0, 1>(const ConvParams);
// the 16 x 16 x 32 form of the deep vocoder convs (round 4)
template __global__ void conv_h16_kernel<128, 128, 2, 2, 1, 1>(const ConvParams);
template __global__ void conv_h16_kernel<128, 128, 2, 2, 3, 1>(const ConvParams);
template __global__ void conv_h16_kernel<64, 128, 2, 2, 1, 1>(const ConvParams);
template __global__ void resblock_pair_h16q_kernel<2, 2, 1>(const PairParams);
template __global__ void resblock_pair_h16q_kernel<1, 4, 1>(const PairParams);
template __global__ void resblock_pair_h16q_kernel<4, 1, 1>(const PairParams);
template __global__ void resblock_pair_h16q_kernel<2, 2, 3>(const PairParams);
template __global__ void resblock_pair_h16q_kernel<1, 4, 3>(const PairParams);
