#!/usr/bin/env python
"""Generates show-edit-tell_amd/csrc/experimental/gemm_gen_asm.inc (an experiment that tied, EXPERIMENTS.md 5.2: built only with
-DSET_EXPERIMENTAL_GEMMS): the hand-scheduled k-loop of the 128x64 general-layout fp32 GEMM
(gemm_gen.hip `gemm_gen_asm<A_KMAJ>`) as one inline-asm string per operand layout.

    python tools/gen_gemm_gen_asm.py        # rewrites the .inc (committed; the build does not run this script)

Why generated: the two layouts (A k-minor = weight gradients dW = dY^T X, A k-major = input gradients dX = dY W; B is k-minor
in both) differ only in how A's LDS image is written and read, and every LDS offset is an immediate that depends on
(k-block, LDS buffer, sub-tile) — 100 literals that are computed here instead of typed.

Tile 128x64x32, 4 waves (2 x 2), wave = 2 accumulators of 32x32 (rows wm*64 + {0, 32}, columns wn*32).  LDS per buffer:
A 16 KB at +0, B 8 KB at +16384; buffer 1 at +24576.  Register map (fixed; the asm statement clobbers v48-v119):
    v48-v63  stage 0, A (4 x 16 B)    v64-v71  stage 0, B (2 x 16 B)
    v72-v87  stage 1, A               v88-v95  stage 1, B
    v96-v107  fragment set X (A sub-tile 0, A sub-tile 1, B: 4 k each)     v108-v119  fragment set Y
Pipeline = gemm_nt_f32_asm's (gemm_f32.hip): two register stages run two k-tiles ahead, the older one is waited for with
vmcnt(6) — exactly the younger stage's six requests may still fly —, one barrier per k-tile, requests past the slice are not
issued (scalar branches on the k-tiles left)."""
import os

BUF1 = 24576            # bytes
BOFF = 16384            # B's image inside a buffer


def frag(kk, buf, dst, a_kmaj):
    """LDS -> fragment set `dst` (base register number) for k-block kk of the tile in buffer `buf`"""
    a0, a1, b = dst, dst + 4, dst + 8
    out = []
    if a_kmaj:
        off = buf * BUF1
        out.append('"ds_read_b128 v[%d:%d], %%[rdA%d] offset:%d\\n"' % (a0, a0 + 3, kk, off))
        out.append('"ds_read_b128 v[%d:%d], %%[rdA%d] offset:%d\\n"' % (a1, a1 + 3, kk, off + 4096))
    else:
        u = buf * (BUF1 // 256) + kk * 16          # units of 256 B: k-row = 512 B, k-block = 8 rows
        for reg, base in ((a0, "rdA0"), (a1, "rdA1")):
            out.append('"ds_read2st64_b32 v[%d:%d], %%[%s] offset0:%d offset1:%d\\n"' % (reg, reg + 1, base, u, u + 2))
            out.append('"ds_read2st64_b32 v[%d:%d], %%[%s] offset0:%d offset1:%d\\n"' % (reg + 2, reg + 3, base, u + 4, u + 6))
    u = buf * (BUF1 // 256) + kk * 8               # B: k-row = 256 B (rdB already carries +16384)
    out.append('"ds_read2st64_b32 v[%d:%d], %%[rdB] offset0:%d offset1:%d\\n"' % (b, b + 1, u, u + 1))
    out.append('"ds_read2st64_b32 v[%d:%d], %%[rdB] offset0:%d offset1:%d\\n"' % (b + 2, b + 3, u + 2, u + 3))
    return out


def mfma(src):
    a0, a1, b = src, src + 4, src + 8
    out = []
    for e in range(4):
        out.append('"v_mfma_f32_32x32x2_f32 %%[acc0], v%d, v%d, %%[acc0]\\n"' % (a0 + e, b + e))
        out.append('"v_mfma_f32_32x32x2_f32 %%[acc1], v%d, v%d, %%[acc1]\\n"' % (a1 + e, b + e))
    return out


def loads(stage):
    base = 48 + 24 * stage
    out = []
    for i in range(4):
        out.append('"buffer_load_dwordx4 v[%d:%d], %%[va%d], %%[rsA], 0 offen\\n"' % (base + 4 * i, base + 4 * i + 3, i))
    for i in range(2):
        out.append('"buffer_load_dwordx4 v[%d:%d], %%[vb%d], %%[rsB], 0 offen\\n"' % (base + 16 + 4 * i, base + 19 + 4 * i, i))
    for i in range(4):
        out.append('"v_add_u32 %%[va%d], %%[stepA], %%[va%d]\\n"' % (i, i))
    for i in range(2):
        out.append('"v_add_u32 %%[vb%d], %%[stepB], %%[vb%d]\\n"' % (i, i))
    return out


def stores(stage, buf):
    base = 48 + 24 * stage
    off = buf * BUF1
    out = []
    for i in range(4):
        out.append('"ds_write_b128 %%[wrA], v[%d:%d] offset:%d\\n"' % (base + 4 * i, base + 4 * i + 3, off + 4096 * i))
    for i in range(2):
        out.append('"ds_write_b128 %%[wrB], v[%d:%d] offset:%d\\n"' % (base + 16 + 4 * i, base + 19 + 4 * i, off + 4096 * i))
    return out


def half(buf, stage, lbl, a_kmaj):
    """one k-tile: MFMAs from buffer `buf`; register stage `stage` (tile kt + 1) -> the other buffer, then refilled (tile kt + 3)"""
    n = 4 if a_kmaj else 6                    # LDS instructions of one fragment set
    X, Y = 96, 108
    o = []
    o += frag(0, buf, X, a_kmaj) + frag(1, buf, Y, a_kmaj)
    o.append('"s_waitcnt lgkmcnt(%d)\\n"' % n)
    o += mfma(X)
    o.append('"s_cmp_lt_i32 %[rem], 2\\n"')
    o.append('"s_cbranch_scc1 %s_nostore%%=\\n"' % lbl)
    o.append('"s_cmp_gt_i32 %[rem], 2\\n"')
    o.append('"s_cbranch_scc1 %s_w6%%=\\n"' % lbl)
    o.append('"s_waitcnt vmcnt(0)\\n"')
    o.append('"s_branch %s_wd%%=\\n"' % lbl)
    o.append('"%s_w6%%=:\\n"' % lbl)
    o.append('"s_waitcnt vmcnt(6)\\n"')
    o.append('"%s_wd%%=:\\n"' % lbl)
    o += stores(stage, buf ^ 1)
    o.append('"s_cmp_lt_i32 %[rem], 4\\n"')
    o.append('"s_cbranch_scc1 %s_nostore%%=\\n"' % lbl)
    o += loads(stage)
    o.append('"%s_nostore%%=:\\n"' % lbl)
    o += frag(2, buf, X, a_kmaj)
    o.append('"s_waitcnt lgkmcnt(%d)\\n"' % n)       # (LDS operations retire in order: all but the n newest)
    o += mfma(Y)
    o += frag(3, buf, Y, a_kmaj)
    o.append('"s_waitcnt lgkmcnt(%d)\\n"' % n)
    o += mfma(X)
    o.append('"s_waitcnt lgkmcnt(0)\\n"')
    o.append('"s_barrier\\n"')
    o += mfma(Y)
    return o


def body(a_kmaj):
    o = []
    # ---- prologue: tile 0 -> stage 0 -> buffer 0; tile 1 -> stage 1; tile 2 -> stage 0
    o += loads(0)
    o.append('"s_cmp_lt_i32 %[rem], 2\\n"')
    o.append('"s_cbranch_scc1 P_one%=\\n"')
    o += loads(1)
    o.append('"s_waitcnt vmcnt(6)\\n"')
    o.append('"s_branch P_st%=\\n"')
    o.append('"P_one%=:\\n"')
    o.append('"s_waitcnt vmcnt(0)\\n"')
    o.append('"P_st%=:\\n"')
    o += stores(0, 0)
    o.append('"s_cmp_lt_i32 %[rem], 3\\n"')
    o.append('"s_cbranch_scc1 P_go%=\\n"')
    o += loads(0)
    o.append('"P_go%=:\\n"')
    o.append('"s_waitcnt lgkmcnt(0)\\n"')
    o.append('"s_barrier\\n"')
    o.append('"L_top%=:\\n"')
    o += half(0, 1, "A", a_kmaj)
    o.append('"s_sub_i32 %[rem], %[rem], 1\\n"')
    o.append('"s_cmp_eq_u32 %[rem], 0\\n"')
    o.append('"s_cbranch_scc1 L_end%=\\n"')
    o += half(1, 0, "B", a_kmaj)
    o.append('"s_sub_i32 %[rem], %[rem], 1\\n"')
    o.append('"s_cmp_eq_u32 %[rem], 0\\n"')
    o.append('"s_cbranch_scc0 L_top%=\\n"')
    o.append('"L_end%=:\\n"')
    # the compiler does not know that the statement ends in MFMAs (see gemm_nt_f32_asm)
    o.append('"s_nop 15\\n"')
    o.append('"s_nop 7\\n"')
    return o


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "show-edit-tell_amd", "csrc", "experimental", "gemm_gen_asm.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_gen_asm.py — do not edit.  The k-loop of gemm_gen_asm<A_KMAJ> (gemm_gen.hip) as inline asm.\n")
        for name, kmaj in (("GEN_ASM_BODY_AKMIN", False), ("GEN_ASM_BODY_AKMAJ", True)):
            lines = body(kmaj)
            f.write("#define %s \\\n" % name)
            f.write(" \\\n".join("    " + ln for ln in lines))
            f.write("\n\n")
        f.write("#define GEN_ASM_CLOBBERS " + ", ".join('"v%d"' % r for r in range(48, 120)) + "\n")
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
