#!/usr/bin/env python3
"""Budget of a fused Winograd F(2x2, 3x3) form of the planar MFMA convolution on one MI355X CU, from rates MEASURED on this board (no GPU needed).

Question (round-5 verdict, item 1): Winograd issues 16 instead of 36 products per 2x2 outputs -- can a fused kernel (input transform in the CU, 16
[tiles x C] x [C x O] products, output transform in the epilogue) turn that into time?  The 16 products share no operand, so what one workgroup can
keep resident decides how often every weight and every input pixel crosses L2 -> LDS; both limits are fixed numbers:

  register file        512 KB per CU = 131 072 fp32 words; conv_planar_kernel keeps 65 536 of them as accumulators (256 px x 128 ch x 2 sets)
  L2 -> LDS staging    32 B/clk/CU sustained (conv_kxr_kernel's producer waves alone: 1 660 cycles per 53-KB stage, DESIGN.md section 4 "Narrow
                       layers"; the guide's 16.8-18.8 TB/s aggregate is the same number x 256 CUs x 2.1 GHz)
  matrix pipe          one v_mfma_f32_16x16x32_f16 = 16 clocks of one SIMD, four SIMDs per CU; three plane products per fp32 product

Per 32-channel K-slab a workgroup with Mt tiles (2x2 output pixels each), Nt output channels and Pc of the 16 transform positions accumulating at once
needs (Pc + 4 if Pc < 16) * Mt * Nt accumulator words (the + 4: the 2x2 outputs being assembled across position passes),
runs Pc * (Mt/16) * (Nt/16) * 3 MFMAs = Pc * Mt * Nt * 3 / 64 clocks, and must stage Pc * Nt * 32 channels * 2 B * 3 planes of transformed weights
(one accumulator set needs the weights as w_h, w_l and w_h / 2048 so that all three products land at unit scale; with two planes and a second
accumulator set the tile halves instead).  Required weight rate = 4096 / Mt B/clk at the full matrix rate -- independent of Nt and Pc.
"""
ACC_WORDS = 65536
L2_LDS_BPC = 32.0
MFMA_FRAC = 0.58          # what the MFMA-bound launches of the step sustain on this board (BENCH_r05: 0.58-0.60)


def row(Pc, Mt, Nt, cout=256):
    acc = (Pc + (4 if Pc < 16 else 0)) * Mt * Nt
    clk = Pc * Mt * Nt * 3 / 64.0
    wbytes = Pc * Nt * 32 * 2 * 3
    passes = (16 // Pc) * (cout // Nt)          # how often the tile's input patch is staged + transformed per K-slab
    # raw input patch of Mt tiles (2 planes of fp16, halo ~1.27x) per pass, over the clocks of ONE pass
    xbytes = Mt * 4 * 1.27 * 32 * 2 * 2
    # the transform: per input element of the staged patch ~24 VALU operations per position pass share (plane join, B^T d B: 32 adds per 16 values,
    # split into two fp16 planes: ~3 per value); one wave instruction = 64 lanes = 4 clocks of one SIMD; v_pk_* forms halve it at best
    valu_clk = Mt * 4 * 1.27 * 32 * 24 * (Pc / 16.0) / 64 * 4 / 4
    return dict(Pc=Pc, Mt=Mt, Nt=Nt, acc=acc, fits=acc <= ACC_WORDS, w_rate=wbytes / clk, x_rate=xbytes / clk, passes=passes,
                total_rate=(wbytes + xbytes) / clk, valu=valu_clk / clk)


def main():
    print(__doc__)
    print("existing conv_planar_kx3_kernel, 256 px x 128 ch, per K-slab and tap: 1 536 MFMA clocks, weights 16 KB = 10.7 B/clk, activations (kx reuse) 7 B/clk"
          " -> 17.7 B/clk at the full matrix rate, %.1f at the sustained %.2f\n" % (17.7 * MFMA_FRAC, MFMA_FRAC))
    print("%4s %5s %5s %8s %5s | %12s %12s %12s | %10s | %s" % ("Pc", "Mt", "Nt", "acc", "fits", "weights B/clk", "input B/clk", "total B/clk", "VALU/MFMA",
                                                               "input passes per K-slab (256 out channels)"))
    for Pc, Mt, Nt in ((16, 32, 128), (16, 64, 64), (16, 128, 32), (16, 256, 16), (8, 64, 64), (4, 64, 128), (4, 128, 64), (4, 256, 32), (1, 128, 64), (1, 256, 32)):
        r = row(Pc, Mt, Nt)
        print("%4d %5d %5d %8d %5s | %12.1f %12.1f %12.1f | %10.2f | %d" % (r["Pc"], r["Mt"], r["Nt"], r["acc"], "yes" if r["fits"] else "NO", r["w_rate"], r["x_rate"],
                                                                      r["total_rate"], r["valu"], r["passes"]))
    best = min((row(*c) for c in ((16, 32, 128), (16, 64, 64), (16, 128, 32), (16, 256, 16), (4, 64, 128), (4, 128, 64))), key=lambda r: r["total_rate"])
    print("\nBest shape: Pc %d, Mt %d, Nt %d: %.0f B/clk at the full matrix rate, %.0f at the sustained %.2f -- above the %.0f B/clk a CU can stage and %.1fx the direct\n"
          "kernel's %.1f.  In bytes per output (0.44x the clocks): %.2fx the direct kernel's L2 -> LDS traffic, with the input transformed %d times per K-slab by VALU\n"
          "instructions the direct kernel does not have at all -- VALU / MFMA = transform clocks over matrix clocks of the same SIMDs (an MFMA-issuing wave leaves\n"
          "a co-resident wave's VALU stream the leftovers: DESIGN.md section 4 item 1): %.2f here; the shapes under 1.0 are the ones that need 64-128 B/clk of weights."
          % (best["Pc"], best["Mt"], best["Nt"], best["total_rate"], best["total_rate"] * MFMA_FRAC, MFMA_FRAC, L2_LDS_BPC, best["total_rate"] / 17.7, 17.7 * MFMA_FRAC,
             best["total_rate"] * (16 / 36) / 17.7, best["passes"], best["valu"]))
    # energy at the power cap (per launch of the 145-GF proto layer MFMAs 1.23 J, activation staging 0.40 J, weight staging 0.25 J)
    e_mfma, e_stage = 1.23, 0.65
    e_w = e_mfma * 16 / 36 + e_stage * best["total_rate"] * (16 / 36) / 17.7
    print("Energy per launch of the 145-GF layer at the 1.4-kW cap (measured split of the direct kernel: MFMAs %.2f J + staging %.2f J = %.2f J): Winograd MFMAs %.2f J +\n"
          "staging %.2f J = %.2f J BEFORE its transforms (%d x 24 VALU operations per input element and K-slab) -- no gain where the >= 1.3x the verdict asks for would\n"
          "need <= %.2f J.  Not built." % (e_mfma, e_stage, e_mfma + e_stage, e_mfma * 16 / 36, e_w - e_mfma * 16 / 36, e_w, best["passes"], (e_mfma + e_stage) / 1.3))


if __name__ == "__main__":
    main()
