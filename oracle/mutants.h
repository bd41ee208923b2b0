/*
 * mutants.h — one identifier per quirk of SURVEY.md §8.1 that the mutation build of the oracle
 * (make -C oracle mutants -> _build/liboracle_mut.so, -DORACLE_MUTANTS) can "fix" at run time with
 * oracle_set_mutant(id). TEST INFRASTRUCTURE ONLY: tests/test_mutants_cpu.py shows each mutant differs
 * from the faithful restatement on the committed fixtures, tests/test_mutants_gpu.py that the HIP path
 * agrees with the faithful build and disagrees with every mutant. The faithful liboracle.so contains
 * none of this (MUT(x) == 0 at compile time). Citations: /root/reference/src/... lines of the quirk.
 */
#ifndef DSDTM_ORACLE_MUTANTS_H
#define DSDTM_ORACLE_MUTANTS_H

enum oracle_mutant_id {
    MUT_NONE = 0,
    /* sparse alignment — src/Sprase_ImageAlign.cpp */
    MUT_Q1_FX = 1,        /* :70,160  Jacobian scale from fx instead of the single focal Camera.f          */
    MUT_Q3_TIGHT = 2,     /* :245,262 current-side border 2 (the footprint) instead of mnboarder = 3        */
    MUT_Q3_ZTEST = 3,     /* :254-262 a z > 0 visibility test the reference does not have                    */
    MUT_Q3_INITIAL = 4,   /* :86      features with !mbInitial not skipped                                   */
    MUT_Q3_ZERO = 5,      /* :93-100  map points that are exactly zero not skipped                           */
    MUT_Q4_SHIFT = 6,     /* :143,278 patch spans -1..+2 instead of -2..+1 from floor(px)                    */
    MUT_Q5_RAWGRAD = 7,   /* :150-158 gradients of the raw pixels instead of the interpolated image          */
    MUT_Q6_TREF = 8,      /* :117-119 X = T_ref * P_w instead of bearing * |P_w - C_ref|                     */
    MUT_Q8_LEFT = 9,      /* :335     T <- exp(x) * T instead of T * exp(x)                                  */
    MUT_Q9_GE = 10,       /* :328     revert on chi2New >= chi2 instead of >                                 */
    MUT_Q9_SUM = 11,      /* :298     chi2 not divided by the visible pixel count                            */
    MUT_Q9_EPS = 12,      /* :305,341 exit at max|x| <= 1e-6 instead of 1e-8                                 */
    MUT_Q10_COARSE = 13,  /* :59      returned count taken at the coarsest level instead of the finest       */
    MUT_Q10_MINFTS = 14,  /* :34      Min_fts compared with the initialised features only                    */
    MUT_Q11_ZERO = 15,    /* :298     chi2 = 0 instead of 0/0 = NaN when no patch is visible                         */
    /* Align2D — src/Feature_alignment.cpp:318-417 */
    MUT_A1_DOUBLE = 20,   /* :330-398 double arithmetic instead of float                                     */
    MUT_A1_ROWSUMS = 21,  /* :386-392 Jres summed row by row and then over rows, not in one raster-order chain    */
    MUT_A3_STRICT = 22,   /* :367-368 u_r == cols-4 / v_r == rows-4 rejected                                 */
    MUT_A4_NOWRITE = 23,  /* :414     px not written back when the alignment fails                           */
    MUT_A4_NOMEAN = 24,   /* :386     no mean-offset term in the residual                                    */
    MUT_A2_CHECK = 25,    /* :345     a conditioning check on H the reference does not have (NaN is what it returns) */
    /* warp prelude — src/Feature_alignment.cpp:160-275 */
    MUT_W1_FLOATDIV = 30, /* :231     1.0f/(1<<level) instead of the integer division                        */
    MUT_W2_ROUND = 31,    /* :254     rounding instead of truncation to u8                                   */
    MUT_W2_REFLEVEL = 32, /* :215-216 reference pixel not divided by its level's scale                       */
    MUT_A13_DET = 33,     /* :198     search level raised at det > 2 instead of det > 3                      */
    /* pose-only refinement — src/Optimizer.cpp:20-101, include/Optimizer.h:129-250 (pose_opt_oracle.c) */
    MUT_P1_JSCALE = 40,   /* h:162,176 the Jacobian divided by 1 << level like the residual (the reference does not)  */
    MUT_P2_NOLOSS = 41,   /* cpp:33   no robust loss instead of CauchyLoss(1.0)                                      */
    MUT_P3_PLUS = 42,     /* h:222-236 Plus as a vector sum instead of SE3(delta) * SE3(x)                            */
    MUT_P4_ALLFEAT = 43,  /* cpp:47-65 residual blocks for every feature, not only Mpt && !IsBad && mbInitial         */
    MUT_P5_PIXELS = 44,   /* h:160    observation taken as bearing.xy (not divided by bearing.z)                      */
    MUT_P6_ITERS = 45,    /* cpp:70   max_num_iterations 10 instead of 100                                           */
    /* cv::pyrDown (src/Frame.cpp:74-81) and Feature_detector::detect (src/Feature_detection.cpp:69-198, Thirdparty/fast) */
    MUT_PD_ROUND = 50,    /* pyr      (s + 127) >> 8 instead of (s + 128) >> 8                                       */
    MUT_PD_BORDER = 51,   /* pyr      BORDER_REFLECT (edge pixel repeated) instead of BORDER_REFLECT_101             */
    MUT_D_NMS_TIE = 52,   /* nonmax_3x3.cpp:47-106 a corner suppressed only by a STRICTLY greater neighbour           */
    MUT_D_SCORE = 53,     /* fast_10_score.cpp the score is the margin itself, not margin - 1                          */
    MUT_D_CELLMAX = 54,   /* Feature_detection.cpp:104 >= instead of > (the LAST of equal scores wins the cell)        */
    MUT_D_BOX = 55        /* Feature_detection.cpp:173-185 Shi-Tomasi box -4..+4 (9x9) instead of -4..+3 (8x8)         */
};

#ifdef ORACLE_MUTANTS
#ifdef __cplusplus
extern "C" {
#endif
extern int oracle_mutant_id;            /* defined in dsdtm_oracle.c */
void oracle_set_mutant(int id);
int oracle_get_mutant(void);
#ifdef __cplusplus
}
#endif
#define MUT(id) (oracle_mutant_id == (id))
#else
#define MUT(id) 0
#endif
#endif
