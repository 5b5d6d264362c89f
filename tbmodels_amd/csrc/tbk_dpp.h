// tbk_dpp.h -- f64 FMAs with a DPP row_newbcast operand (gfx90a and later; device code only).
//
// acc += x * (lane T of `src` in this lane's row of 16 lanes)   /   acc -= ...
// The f64 FMA is the one f64 arithmetic instruction with a VOP2 encoding here (v_fmac_f64), so it takes a DPP source;
// `row_newbcast:T` hands every lane of a row the value lane T of that row holds (tools/dpp_probe.hip: right lanes, the
// neg modifier works, 93 % of the plain FMA rate).  A value all lanes need in the same instruction -- v[c], w[c] of a
// Householder reflector for the column c a register holds, a pending panel row, the 8 x 8 factors of the compact WY
// form -- then comes out of ONE register whose lane t holds the t-th value, instead of an LDS broadcast read per value
// (a 64-lane read that moves 1 KiB through the LDS pipe to deliver 16 bytes).
//
// Hazards.  The DPP FMAs go out through inline asm, and LLVM's hazard recogniser does not look inside asm blocks:
// gfx9 needs 2 wait states between a VALU write of the VGPR a DPP operand reads and the DPP instruction, and 5 after a
// VALU write of EXEC; the hardware does not interlock them.  An `s_nop 1` in front of every DPP FMA was measured:
// +10 % on the n <= 64 reduction (12 such FMAs per column and step), so it is not there.  Instead the BUILT objects
// are checked: tools/dpp_hazard_lint.py disassembles every kernel and fails (tests/test_dpp_hazards.py, CPU suite)
// if a VALU instruction writes a DPP source register less than 2 instructions, or EXEC less than 5 instructions,
// ahead of a v_fmac_f64_dpp -- a scheduler change that creates the hazard breaks the build check, not the numbers.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

template <int T>
__device__ __forceinline__ void fmac_bc(double& acc, double src, double x) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(x), "n"(T));
}
template <int T>
__device__ __forceinline__ void fnmac_bc(double& acc, double src, double x) {
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(x), "n"(T));
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
