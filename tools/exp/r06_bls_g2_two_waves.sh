#!/bin/bash
# round 6: could the BLS12-381 G2 bucket kernel live at TWO waves per SIMD?  Compile-only experiment (no GPU): the same kernel with the
# occupancy FORCED to two waves per SIMD (amdgpu_waves_per_eu(2,2): 256 of the 512 unified registers per lane) -- with the WHOLE accumulator
# still in LDS, i.e. before the 56 registers ZZ, ZZZ would take on top.  (The kernel already ASKS for two waves per SIMD --
# __launch_bounds__'s second argument is waves per SIMD in HIP -- and the compiler answers "failed to meet occupancy target": it
# will not spill to get there on its own.)  Prints registers, scratch and spills.
cd "$(dirname "$0")/../../gnark_amd/csrc"
mkdir -p build_exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -Wno-unused-result -Wno-unused-value -ffp-contract=off \
  -DGA_ACC29_FP2_MINW=2 -DGA_ACC29_NUM_VGPR=2 -shared msm_bls12381_g2.hip -o build_exp/msm_bls12381_g2_minw4.so 2> build_exp/minw4.log
python ../../tools/kernel_resources.py "msm_accumulate29_kernel<ga::Fe2<ga::BLS12_381_Fp>" --so build_exp/msm_bls12381_g2_minw4.so
python ../../tools/isa_count.py "msm_accumulate29_kernel<ga::Fe2<ga::BLS12_381_Fp>, false" --so build_exp/msm_bls12381_g2_minw4.so 2>/dev/null | cut -c1-260
