#!/usr/bin/env python3
"""Writes deq-mpc-corl_amd/csrc/alqp_ipm_g4_gpu.hpp: the gfx950 execution policy of alqp_ipm_g4.hpp.

The row-broadcast FMA chains (v_fmac_{f32,f64}_dpp row_newbcast) are inline asm, one block per chain length and
dtype; they are mechanical repetitions, so they are generated. Run from the repo root:  python tools/gen_g4_policy.py
"""
import os

MAXN = 18
SFX = {"float": "f32", "double": "f64"}
MOVZ = {"float": "v_mov_b32 %{t}, 0", "double": "v_mov_b64 %{t}, 0"}
TAIL = " row_mask:0xf bank_mask:0xf\\n\\t"


def row(n, t):
    """acc += sum_i bcast_{K0+i}(x) * m[i]. ONE accumulator: measured on gfx950 (tools/probes/dp_latency_probe.hip) a
    dependent v_fmac_f64_dpp chain issues every 4.3 cycles, the same as independent ones (two alternating accumulators:
    5.2 per instruction plus a move and an add)."""
    two = False
    lines = ['"s_nop 1\\n\\t"']
    if two:
        lines.append('"' + MOVZ[t].format(t=1) + '\\n\\t"')
    base = 2 if two else 1          # %0 acc, (%1 tmp), then x, m[0..n), K0..
    xi, mi, ki = base, base + 1, base + 1 + n
    for i in range(n):
        dst = 1 if (two and i % 2 == 1) else 0
        lines.append(f'"v_fmac_{SFX[t]}_dpp %{dst}, %{xi}, %{mi + i} row_newbcast:%{ki}+{i}{TAIL}"')
    outs = '"+&v"(acc)' + (', "=&v"(tmp)' if two else "")
    ins = ", ".join(['"v"(x)'] + [f'"v"(m[{i}])' for i in range(n)] + ['"n"(K0)'])
    body = "        asm(" + "\n            ".join(lines) + f"\n            : {outs} : {ins});\n"
    if two:
        body = f"        {t} tmp;\n" + body + "        acc += tmp;\n"
    return f"    template <int K0> static G4_FN void row_{n}({t} &acc, const {t} &x, const {t} *m) {{\n{body}    }}\n"


def multi(n, t):
    """acc[i] += bcast_{K0+i}(x) * m  (independent accumulators)."""
    lines = ['"s_nop 1\\n\\t"']
    for i in range(n):
        lines.append(f'"v_fmac_{SFX[t]}_dpp %{i}, %{n}, %{n + 1} row_newbcast:%{n + 2}+{i}{TAIL}"')
    outs = ", ".join(f'"+&v"(acc[{i}])' for i in range(n))
    ins = ", ".join(['"v"(x)', '"v"(m)', '"n"(K0)'])
    return (f"    template <int K0> static G4_FN void multi_{n}({t} *acc, const {t} &x, const {t} &m) {{\n"
            "        asm(" + "\n            ".join(lines) + f"\n            : {outs} : {ins});\n    }}\n")


def selfu(n, t):
    """acc[i] += bcast_K(acc[i]) * m  (every register is updated with lane K's value of ITSELF: the row operation
    row_r += m_r * row_K of a matrix held one row per lane)."""
    lines = ['"s_nop 1\\n\\t"']
    for i in range(n):
        lines.append(f'"v_fmac_{SFX[t]}_dpp %{i}, %{i}, %{n} row_newbcast:%{n + 1}{TAIL}"')
    outs = ", ".join(f'"+&v"(acc[{i}])' for i in range(n))
    return (f"    template <int K> static G4_FN void self_{n}({t} *acc, const {t} &m) {{\n"
            "        asm(" + "\n            ".join(lines) + f"\n            : {outs} : \"v\"(m), \"n\"(K));\n    }}\n")


def vec(n, t):
    """acc += sum_i bcast_K(x[i]) * m[i]; one accumulator (as row)."""
    two = False
    lines = ['"s_nop 1\\n\\t"']
    if two:
        lines.append('"' + MOVZ[t].format(t=1) + '\\n\\t"')
    base = 2 if two else 1
    xi, mi, ki = base, base + n, base + 2 * n
    for i in range(n):
        dst = 1 if (two and i % 2 == 1) else 0
        lines.append(f'"v_fmac_{SFX[t]}_dpp %{dst}, %{xi + i}, %{mi + i} row_newbcast:%{ki}{TAIL}"')
    outs = '"+&v"(acc)' + (', "=&v"(tmp)' if two else "")
    ins = ", ".join([f'"v"(x[{i}])' for i in range(n)] + [f'"v"(m[{i}])' for i in range(n)] + ['"n"(K)'])
    body = "        asm(" + "\n            ".join(lines) + f"\n            : {outs} : {ins});\n"
    if two:
        body = f"        {t} tmp;\n" + body + "        acc += tmp;\n"
    return f"    template <int K> static G4_FN void vec_{n}({t} &acc, const {t} *x, const {t} *m) {{\n{body}    }}\n"


NXS = (2, 4, 6, 8, 10, 12, 13, 14)   # state sizes of csrc/alqp_dims.hpp: the whole-matrix blocks below exist for these


def rank(na, nt, t):
    """acc[i] += bcast_i(x[k]) * m[k], i < na, for k < nt in turn: nt rank-1 updates of an na-column row set in ONE
    block (one hazard pad instead of nt; na + 2 nt operands <= 30)."""
    lines = ['"s_nop 1\\n\\t"']
    for k in range(nt):
        for i in range(na):
            lines.append(f'"v_fmac_{SFX[t]}_dpp %{i}, %{na + k}, %{na + nt + k} row_newbcast:{i}{TAIL}"')
    outs = ", ".join(f'"+&v"(acc[{i}])' for i in range(na))
    ins = ", ".join([f'"v"(x[{k}])' for k in range(nt)] + [f'"v"(m[{k}])' for k in range(nt)])
    return (f"    static G4_FN void rank_{na}_{nt}({t} *acc, const {t} *x, const {t} *m) {{\n"
            "        asm(" + "\n            ".join(lines) + f"\n            : {outs} : {ins});\n    }}\n")


def pivot(nx, c, t):
    """One elimination step of the L D L' / M = L^-1 pivot loop as ONE block: the trailing columns of S
    (S[k] += bcast_k(S[c]) * nl, k > c) and the leading columns of M (M[i] += bcast_c(M[i]) * nl, i < c)."""
    n1, n2 = nx - 1 - c, c
    if n1 + n2 == 0:
        return ""
    lines = ['"s_nop 1\\n\\t"']
    xo, mo = n1 + n2, n1 + n2 + 1
    for i in range(n1):
        lines.append(f'"v_fmac_{SFX[t]}_dpp %{i}, %{xo}, %{mo} row_newbcast:{c + 1 + i}{TAIL}"')
    for i in range(n2):
        lines.append(f'"v_fmac_{SFX[t]}_dpp %{n1 + i}, %{n1 + i}, %{mo} row_newbcast:{c}{TAIL}"')
    outs = ", ".join([f'"+&v"(S[{c + 1 + i}])' for i in range(n1)] + [f'"+&v"(M[{i}])' for i in range(n2)])
    return (f"    static G4_FN void pivot_{nx}_{c}({t} *S, {t} *M, const {t} &nl) {{\n"
            "        asm(" + "\n            ".join(lines) + f"\n            : {outs} : \"v\"(S[{c}]), \"v\"(nl));\n    }}\n")


def scale(n, t):
    """out[i] = bcast_{K0+i}(x) * m[i]: a vector held one element per lane scales n registers (f64: zero + fmac, the
    double-precision ALU has no DPP multiply; f32: v_mul_f32_dpp)."""
    lines = ['"s_nop 1\\n\\t"']
    for i in range(n):
        if t == "double":
            lines.append(f'"v_mov_b64 %{i}, 0\\n\\t"')
            lines.append(f'"v_fmac_f64_dpp %{i}, %{n}, %{n + 1 + i} row_newbcast:%{2 * n + 1}+{i}{TAIL}"')
        else:
            lines.append(f'"v_mul_f32_dpp %{i}, %{n}, %{n + 1 + i} row_newbcast:%{2 * n + 1}+{i}{TAIL}"')
    outs = ", ".join(f'"=&v"(out[{i}])' for i in range(n))
    ins = ", ".join(['"v"(x)'] + [f'"v"(m[{i}])' for i in range(n)] + ['"n"(K0)'])
    return (f"    template <int K0> static G4_FN void scale_{n}({t} *out, const {t} &x, const {t} *m) {{\n"
            "        asm(" + "\n            ".join(lines) + f"\n            : {outs} : {ins});\n    }}\n")


def ztri(nx, k0, k1, t):
    """Z[c] += bcast_c(M[k]) * s[k] for k0 <= k < k1, c > k: the strictly-lower part of Z = S M' in two blocks."""
    lo = k0 + 1                       # accumulators Z[lo .. nx)
    na, nk = nx - lo, k1 - k0
    lines = ['"s_nop 1\\n\\t"']
    for k in range(k0, k1):
        for c in range(k + 1, nx):
            lines.append(f'"v_fmac_{SFX[t]}_dpp %{c - lo}, %{na + k - k0}, %{na + nk + k - k0} row_newbcast:{c}{TAIL}"')
    outs = ", ".join(f'"+&v"(Z[{c}])' for c in range(lo, nx))
    ins = ", ".join([f'"v"(M[{k}])' for k in range(k0, k1)] + [f'"v"(s[{k}])' for k in range(k0, k1)])
    return (f"    static G4_FN void ztri_{nx}_{k0}({t} *Z, const {t} *M, const {t} *s) {{\n"
            "        asm(" + "\n            ".join(lines) + f"\n            : {outs} : {ins});\n    }}\n")


def ztri_split(nx):
    """[0, h) and [h, nx-1): both blocks within 30 operands."""
    h = (nx - 1 + 1) // 2
    while (nx - 1) + 2 * h > 30:
        h -= 1
    return h


def dispatcher(name, sig, call, maxn, split=None):
    s = f"    template <int K0, int CNT> static G4_FN void {name}({sig}) {{\n"
    for n in range(1, maxn + 1):
        s += f"        {'if' if n == 1 else 'else if'} constexpr (CNT == {n}) {name}_{n}<K0>({call});\n"
    if split:   # longer chains: one full block, then the rest
        s += f"        else if constexpr (CNT > {maxn}) {{ {name}_{maxn}<K0>({call}); {name}<{split[0]}, CNT - {maxn}>({split[1]}); }}\n"
    return s + "    }\n"


HEAD = r'''// alqp_ipm_g4_gpu.hpp - gfx950 execution policy of alqp_ipm_g4.hpp: one lane = one element of every per-lane
// value (V = real, VI = int, VM = bool).  GENERATED by tools/gen_g4_policy.py - edit that, not this file.
//
// The hot primitive is the row-broadcast FMA, acc += lane_K_of_my_16_lane_row(x) * m, ONE instruction on gfx950:
// v_fmac_f32_dpp / v_fmac_f64_dpp with row_newbcast:K (the only DPP control the double-precision ALU accepts).
// hipcc has no builtin that produces the fused form (__builtin_amdgcn_update_dpp + fma stays two instructions in
// fp32 and three in fp64: the DPP combiner does not know row_newbcast), so the chains are inline asm, one block per
// chain (row_N / multi_N / vec_N below). Rules they follow (cdna_hip_programming.md 5.7):
//   * hipcc does not pad hazards of an asm statement: every block opens with `s_nop 1`, the two wait states a DPP
//     read needs after a VALU write of its source (which may be a compiler copy or an accumulator-register reload
//     placed right in front of the block);
//   * accumulators are early-clobber ("+&v" / "=&v"): a block reads x again after its first write, so no input may
//     share a register with an accumulator;
//   * measured (tools/probes/dp_latency_probe.hip, one wavefront per SIMD): a dependent v_fmac_f64_dpp chain issues
//     every 4.3 cycles, as fast as independent ones, so chains use ONE accumulator; `s_nop 1` costs 8 cycles (two
//     FMAs), which is why a chain is one block and not one block per term; v_rcp_f64 16; an LDS read returns after
//     ~60 cycles, a ds_bpermute after ~68;
//   * no other DPP use anywhere: irregular cross-lane moves go through ds_bpermute (builtin; the LDS crossbar has
//     no such hazard), so the compiler never emits a DPP read of a register an asm block has just written;
//   * all 64 lanes are active wherever these run (gfx9 DPP treats an EXEC-disabled source lane as invalid): the
//     kernel's control flow is wave-uniform, lane predicates live in selects and in masked memory accesses.
#pragma once
#include <hip/hip_runtime.h>

#define G4_FN __device__ __forceinline__
#define G4_UNROLL _Pragma("unroll")
#include "alqp_ipm_g4.hpp"

namespace alqp_ipm_g4 {

template <typename real>
struct GpuX {
    using V = real;
    using VI = int;
    using VM = bool;
    static G4_FN VI lane_id() { return (int)threadIdx.x; }
    static G4_FN V splat(real x) { return x; }
    static G4_FN VI splati(int x) { return x; }
    static G4_FN VM never() { return false; }
    static G4_FN V sel(VM m, V a, V b) { return m ? a : b; }
    static G4_FN VI seli(VM m, VI a, VI b) { return m ? a : b; }
    static G4_FN VI mini(VI a, VI b) { return a < b ? a : b; }
    static G4_FN VI maxi(VI a, VI b) { return a > b ? a : b; }
    static G4_FN float absv(float a) { return __builtin_fabsf(a); }
    static G4_FN double absv(double a) { return __builtin_fabs(a); }
    // 1 / a to ~1 ulp: hardware estimate + Newton steps (the IEEE division sequence is three times as long, and
    // the pivots of the block factorisation sit on the critical path)
    static G4_FN float rcp(float a) {
        float r = __builtin_amdgcn_rcpf(a);
        r = __builtin_fmaf(__builtin_fmaf(-a, r, 1.0f), r, r);
        return r;
    }
    static G4_FN double rcp(double a) {
        double r = __builtin_amdgcn_rcp(a);
        r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
        return r;
    }
    static G4_FN float first(float a) {
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)));
    }
    static G4_FN double first(double a) {
        const long long b = __builtin_bit_cast(long long, a);
        const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
    static G4_FN int firsti(int a) { return __builtin_amdgcn_readfirstlane(a); }
    static G4_FN int lanei(int a, int l) { return __builtin_amdgcn_readlane(a, l); }
    template <int K> static G4_FN float bcast(const float &x) {
        float o;
        asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(x), "n"(K));
        return o;
    }
    template <int K> static G4_FN double bcast(const double &x) {
        double o;
        asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(x), "n"(K));
        return o;
    }
    static G4_FN float gather(float x, int src) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, x)));
    }
    static G4_FN double gather(double x, int src) {
        const long long b = __builtin_bit_cast(long long, x);
        const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b & 0xffffffffll));
        const int hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
'''

TAILSRC = r'''    // memory
    static G4_FN V lds_ld(const real *p, VI idx) { return p[idx]; }
    static G4_FN V lds_ldu(const real *p, int idx) { return p[idx]; }
    static G4_FN void lds_st(real *p, VI idx, V v, VM m) { if (m) p[idx] = v; }
    // v[CNT-1] -> p[idx0 + CNT-1] first, ..., v[0] -> p[idx0] last, under ONE lane predicate and in exactly this order (the
    // fences stop hipcc from reordering or pairing the stores: lanes overwrite each other's cells on purpose, see factor())
    template <int CNT> static G4_FN void lds_st_desc(real *p, VI idx0, const V *v, VM m) {
        if (m) {
            _Pragma("unroll") for (int k = CNT - 1; k >= 0; --k) {
                p[idx0 + k] = v[k];
                asm volatile("" ::: "memory");
            }
        }
    }
    // global accesses as uniform base + 32-bit byte offset (the saddr form: no 64-bit address per lane and access)
    static G4_FN V g_ld(const real *p, VI idx, VM m) {
        return m ? *reinterpret_cast<const real *>(reinterpret_cast<const char *>(p) + (unsigned)idx * (unsigned)sizeof(real)) : real(0);
    }
    static G4_FN void g_st(real *p, VI idx, V v, VM m) {
        if (m) *reinterpret_cast<real *>(reinterpret_cast<char *>(p) + (unsigned)idx * (unsigned)sizeof(real)) = v;
    }
    static G4_FN void launder(int &x) { asm volatile("" : "+v"(x)); }
#ifdef ALQP_G4_TIMING
    static G4_FN long long now() { return (long long)__builtin_amdgcn_s_memtime(); }
    static G4_FN void publish_timing(const long long *t);   // alqp_ipm_g4.hip
#endif
    // one wavefront = one workgroup: LDS instructions of a wave execute in order, so a compiler fence is all a
    // write -> read hand-over between lanes needs (no s_barrier, no vmcnt drain)
    static G4_FN void fence() { asm volatile("" ::: "memory"); }
    static G4_FN void gfence() { __syncthreads(); }
    static G4_FN void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
    // Wave reductions on the DPP network, no LDS round trips: an inclusive scan inside each 16-lane row (row_shr 1, 2,
    // 4, 8), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15), then lane 31 into the upper half (row_bcast31):
    // lane 63 holds the result. (These are compiler-visible DPP moves of values the C++ code computed - hazards are
    // padded by hipcc - unlike the asm chains; the shuffle version cost six exposed ds_bpermute round trips each.)
    template <int CTRL, int ROWS, bool ZFILL> static G4_FN float dpp_mov(float old, float src) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                      CTRL, ROWS, 0xf, ZFILL));
    }
    template <int CTRL, int ROWS, bool ZFILL> static G4_FN double dpp_mov(double old, double src) {
        const long long o = __builtin_bit_cast(long long, old), b = __builtin_bit_cast(long long, src);
        const int lo = __builtin_amdgcn_update_dpp((int)(o & 0xffffffffll), (int)(b & 0xffffffffll), CTRL, ROWS, 0xf, ZFILL);
        const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), CTRL, ROWS, 0xf, ZFILL);
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
    static G4_FN float lane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
    static G4_FN double lane63(double v) {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
    // moves inside a 16-lane row on the DPP network: lane l takes lane l + N (rshl) / l - N (rshr) of its row, 0 beyond it
    template <int N> static G4_FN real rshl(real v) { return dpp_mov<0x100 + N, 0xf, true>(real(0), v); }
    template <int N> static G4_FN real rshr(real v) { return dpp_mov<0x110 + N, 0xf, true>(real(0), v); }
    static G4_FN real wave_sum(real v) {
        v += dpp_mov<0x111, 0xf, true>(real(0), v);
        v += dpp_mov<0x112, 0xf, true>(real(0), v);
        v += dpp_mov<0x114, 0xf, true>(real(0), v);
        v += dpp_mov<0x118, 0xf, true>(real(0), v);
        v += dpp_mov<0x142, 0xa, false>(real(0), v);
        v += dpp_mov<0x143, 0xc, false>(real(0), v);
        return lane63(v);
    }
    static G4_FN float min2(float a, float b) { return __builtin_fminf(a, b); }     // v_min: one instruction (a NaN operand is
    static G4_FN double min2(double a, double b) { return __builtin_fmin(a, b); }   // ignored; the callers flag NaNs separately)
    static G4_FN V vmin(V a, V b) { return min2(a, b); }
    static G4_FN real wave_min(real v) {
        v = min2(v, dpp_mov<0x111, 0xf, false>(v, v));
        v = min2(v, dpp_mov<0x112, 0xf, false>(v, v));
        v = min2(v, dpp_mov<0x114, 0xf, false>(v, v));
        v = min2(v, dpp_mov<0x118, 0xf, false>(v, v));
        v = min2(v, dpp_mov<0x142, 0xa, false>(v, v));
        v = min2(v, dpp_mov<0x143, 0xc, false>(v, v));
        return lane63(v);
    }
    static G4_FN bool wave_any(bool m) { return __builtin_amdgcn_ballot_w64(m) != 0; }
    static G4_FN void store4(real *sc, real a, real b, real c, real d) {
        if (threadIdx.x == 0) { sc[0] = a; sc[1] = b; sc[2] = c; sc[3] = d; }
    }
    template <class A>
    static G4_FN void store_scalars(const A &a, int b, real best, real mu, int iter_best, int improved, int info) {
        if (threadIdx.x == 0) {
            if (a.o_resid) a.o_resid[b] = best;
            if (a.o_mu) a.o_mu[b] = mu;
            if (a.o_iter_best) a.o_iter_best[b] = iter_best;
            if (a.o_improved) a.o_improved[b] = improved;
            if (a.o_info && info && a.o_info[b] == 0) a.o_info[b] = info;
        }
    }
};

}  // namespace alqp_ipm_g4
'''


def main():
    body = ""
    for t in ("float", "double"):
        for n in range(1, MAXN + 1):
            body += row(n, t)
        for n in range(1, 17):
            body += multi(n, t) + selfu(n, t)
        for n in range(1, 14):   # 2 accumulators + 2 n inputs + K: the 30-operand limit of an asm statement
            body += vec(n, t)
        for n in range(1, 15):
            body += scale(n, t)
        for nx in NXS:
            for nt in range(1, min(8, (30 - nx) // 2) + 1):
                body += rank(nx, nt, t)
            for c in range(nx):
                body += pivot(nx, c, t)
            if nx > 1:
                h = ztri_split(nx)
                body += ztri(nx, 0, h, t)
                if h < nx - 1:
                    body += ztri(nx, h, nx - 1, t)
    body += "    // whole-matrix blocks (state sizes of alqp_dims.hpp)\n"
    body += "    template <int NA, int NT> static G4_FN void rank(real *acc, const real *x, const real *m) {\n"
    first = True
    for nx in NXS:
        for nt in range(1, min(8, (30 - nx) // 2) + 1):
            body += f"        {'if' if first else 'else if'} constexpr (NA == {nx} && NT == {nt}) rank_{nx}_{nt}(acc, x, m);\n"
            first = False
    body += "        else static_assert(NA < 0, \"rank: no block of this shape\");\n    }\n"
    body += "    static constexpr int rank_max(int na) { return (30 - na) / 2 < 8 ? (30 - na) / 2 : 8; }   // terms per block\n"
    body += "    template <int C, int NXX> static G4_FN void pivot(real *S, real *M, const real &nl) {\n"
    first = True
    for nx in NXS:
        for c in range(nx):
            if nx - 1 - c + c == 0:
                continue
            body += f"        {'if' if first else 'else if'} constexpr (NXX == {nx} && C == {c}) pivot_{nx}_{c}(S, M, nl);\n"
            first = False
    body += "        else static_assert(NXX < 0, \"pivot: no block of this shape\");\n    }\n"
    body += "    template <int NXX> static G4_FN void ztri(real *Z, const real *M, const real *s) {\n"
    first = True
    for nx in NXS:
        h = ztri_split(nx)
        calls = f"ztri_{nx}_0(Z, M, s);" + (f" ztri_{nx}_{h}(Z, M, s);" if h < nx - 1 else "")
        body += f"        {'if' if first else 'else if'} constexpr (NXX == {nx}) {{ {calls} }}\n"
        first = False
    body += "        else static_assert(NXX < 0, \"ztri: no block of this shape\");\n    }\n"
    body += "    // chain-length dispatch (CNT = 0: nothing)\n"
    body += dispatcher("row", "real &acc, const real &x, const real *m", "acc, x, m", MAXN)
    body += dispatcher("multi", "real *acc, const real &x, const real &m", "acc, x, m", 16)
    body += dispatcher("vec", "real &acc, const real *x, const real *m", "acc, x, m", 13, ("K0", "acc, x + 13, m + 13"))
    body += dispatcher("self", "real *acc, const real &m", "acc, m", 16)
    body += dispatcher("scale", "real *out, const real &x, const real *m", "out, x, m", 14)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deq-mpc-corl_amd", "csrc",
                       "alqp_ipm_g4_gpu.hpp")
    with open(out, "w") as f:
        f.write(HEAD + body + TAILSRC)
    print("wrote", out)


if __name__ == "__main__":
    main()
