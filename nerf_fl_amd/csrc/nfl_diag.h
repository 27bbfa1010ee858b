// nfl_diag.h -- every compile-time switch of the DIAGNOSTIC builds (`make diag`, `make variant`), in one place.
// The release library (`make`, __graft_entry__.build()) defines none of them: the #error below refuses a release
// compile that does, so no diagnostic path can reach the product by accident.  A diagnostic library is loaded only
// with NERF_FL_AMD_DEV=1 NFL_LIB=<path> (nerf_fl_amd/_lib.py).
//
//   NFL_STAMPS=1|2                 in-kernel s_memtime phase stamps in the render / dgrad kernel (tests/stamp_*.py)
//   NFL_DIAG_X3_PRODS=0|1|2        nfl_dgrad TU: which correction products the three-product dgrad kernel issues besides
//                                  W_hi d_hi (bit 0: W_lo d_hi, bit 1: W_hi d_lo; nfl_prods.h) -- the gradient CHAIN gets a
//                                  reduced arithmetic while the stashes stay split, i.e. the weight-gradient GEMMs still
//                                  see hi + lo operands
//   NFL_DIAG_WGRAD_PASSES=1|2      nfl_wgrad TU: the split weight gradient stops after d_hi (x) h_hi (1) or after
//                                  adding d_lo (x) h_hi (2) -- single-product weight gradients on a three-product chain
//   NFL_PRODS_OVERRIDE={..16..}    nfl_render_x3 TU: another per-layer product plan of the forward (nfl_prods.h), for the sweep of
//                                  tests/report_prods.py;  NFL_DIAG_INFERENCE_ONLY: compile only the inference instantiation
// The X3_PRODS / WGRAD_PASSES switches attribute an effect of the backward's rounding to the chain or to the weight-gradient products
// (profiles/r03_psnr_backward_attribution.txt).
//   NFL_DIAG_RN_DELTA              nfl_dgrad TU: the single-image dgrad kernels round their gradients to the NEAREST fp16 (rounds 1-2,
//                                  and the attribution file's "f16" / "f16w" rows) instead of stochastically
//   NFL_DIAG_RN_WT                 nfl_pack TU: the transposed weights of the single-image gradient chain are rounded to the NEAREST
//                                  fp16 (rounds 1-2) instead of stochastically
//   NFL_DIAG_WGRAD_LDSDMA          nfl_wgrad TU: timing ablation, the load stream by LDS-DMA into the two LDS slots (nothing consumed or flushed)
//   NFL_DIAG_WGRAD_LOADS_ONLY      nfl_wgrad TU: timing ablation, only the load stream (no LDS round trip, barrier, MFMA): what does the access pattern sustain?
//   WG_SEGMENTS_CONTIGUOUS / WG_D4 .. WG_D8B   nfl_wgrad TU (variant builds): one contiguous segment range per workgroup; segments in flight per wave
//   NFL_DIAG_WGRAD_NOFLUSH         nfl_wgrad TU: timing ablation, the workgroups skip their atomic flush (what does the flush cost?)
#pragma once
#if (defined(NFL_DIAG_WGRAD_NOFLUSH) || defined(NFL_DIAG_WGRAD_LDSDMA) || defined(NFL_DIAG_WGRAD_LOADS_ONLY) || defined(NFL_DIAG_RN_DELTA) || defined(NFL_DIAG_RN_WT) || defined(NFL_STAMPS) || defined(NFL_DIAG_X3_PRODS) || defined(NFL_DIAG_WGRAD_PASSES) || defined(NFL_DIAG_INFERENCE_ONLY)) && !defined(NFL_DIAG_BUILD)
#error "diagnostic switches are only for `make diag` / `make variant` (which pass -DNFL_DIAG_BUILD)"
#endif
