// nfl_prods.h -- the per-layer product plan of the three-product (f16x3) FORWARD.
//
// A layer's matrix product W x with both operands split hi + lo in fp16 is w_hi x_hi + w_lo x_hi + w_hi x_lo
// (+ w_lo x_lo, 2^-22 relative, never issued).  Each correction term costs a third of the layer's MFMA work; what it buys
// depends on the layer: an error made early is carried through every layer behind it, one made in a head reaches the
// output once.  The plan below says which correction terms each layer issues:
//     bit 0: w_lo x_hi  (weights' residuals; dropped = the layer's weights are fp16-rounded)
//     bit 1: w_hi x_lo  (activations' residuals; dropped = the layer's input is fp16-rounded)
// 3 = both (full f16x3), 0 = a single fp16 product.  It is chosen by MEASUREMENT: tests/report_parity.py --prods sweeps
// single-layer drops over all two-pass fixtures and prints each layer's contribution to the worst output error; the plan
// shipped is the cheapest one that keeps every fixture at <= 5e-5 (half the 1e-4 bar of BASELINE.json's north_star).
// Results: profiles/r03_prods_sweep.txt.  The backward is not affected (its stashes hold the operands' images either way).
#pragma once
enum {
    NFL_P_IDX_L1 = 0, NFL_P_IDX_L2, NFL_P_IDX_L3, NFL_P_IDX_L4, NFL_P_IDX_L5, NFL_P_IDX_L6, NFL_P_IDX_L7, NFL_P_IDX_L8,
    NFL_P_IDX_SIG, NFL_P_IDX_DIR, NFL_P_IDX_RGB, NFL_P_IDX_T1, NFL_P_IDX_T2, NFL_P_IDX_T3, NFL_P_IDX_T4, NFL_P_IDX_THEAD,
    NFL_P_IDX_COUNT
};
#ifdef NFL_PRODS_OVERRIDE          // sweep builds only (make variant VFLAGS='-DNFL_PRODS_OVERRIDE={3,3,...}')
#ifndef NFL_DIAG_BUILD
#error "NFL_PRODS_OVERRIDE is a diagnostic switch (make variant)"
#endif
static constexpr int NFL_PRODS[NFL_P_IDX_COUNT] = NFL_PRODS_OVERRIDE;
#else
//                                                  L1 L2 L3 L4 L5 L6 L7 L8 SIG DIR RGB T1 T2 T3 T4 THEAD
static constexpr int NFL_PRODS[NFL_P_IDX_COUNT] = { 3, 3, 3, 3, 3, 3, 3, 3,  3,  3,  3, 3, 3, 3, 3,  3};
#endif
