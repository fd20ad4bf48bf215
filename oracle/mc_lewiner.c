/* Serial CPU restatement of Lewiner marching cubes with scikit-image 0.18.3 semantics.
 * TEST INFRASTRUCTURE ONLY (oracle): the product path never links or loads this file.
 *
 * The reference calls `skimage.measure.marching_cubes_lewiner(sdf, thresh)` (`mesh_util.py:84`), a
 * third-party, un-vendored dependency whose version the reference does not pin; this oracle follows
 * the black-box specification of SURVEY.md Appendix B (verified there against scikit-image 0.18.3)
 * and is pinned by `tests/golden/mc_*.npz`, generated with scikit-image 0.18.3 itself by
 * `tests/golden/make_golden_mc.py`.  Wrapper behaviour reproduced: step_size=1,
 * gradient_direction='descent' (triangle index order reversed), allow_degenerate=True,
 * use_classic=False, vertices returned in array-axis order.
 *
 * Internal axes follow Appendix B item 1: x,y,z = array axes 2,1,0; cells are visited z outer, y, x
 * inner.  Look-up tables: ../rgb-d-pifuhd_amd/csrc/mc_luts.h (BSD data extracted from scikit-image).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MC_LUT_QUAL static const
#include "mc_luts.h"

static const double MC_EPS = 2.220446049250313e-16;   /* np.spacing(1.0): App. B item 4 */

/* edge -> (corner a, corner b) */
static const int EDGE_CORNERS[12][2] = {{0,1},{1,2},{2,3},{3,0},{4,5},{5,6},{6,7},{7,4},{0,4},{1,5},{2,6},{3,7}};
/* corner -> (dx,dy,dz) */
static const int CORNER_OFF[8][3] = {{0,0,0},{1,0,0},{1,1,0},{0,1,0},{0,0,1},{1,0,1},{1,1,1},{0,1,1}};

static int test_face(const double *c, int f)
{
    int A, B, C, D;
    switch (abs(f)) {
    case 1: A = 0; B = 4; C = 5; D = 1; break;
    case 2: A = 1; B = 5; C = 6; D = 2; break;
    case 3: A = 2; B = 6; C = 7; D = 3; break;
    case 4: A = 3; B = 7; C = 4; D = 0; break;
    case 5: A = 0; B = 3; C = 2; D = 1; break;
    default: A = 4; B = 7; C = 6; D = 5; break;
    }
    double d = c[A] * c[C] - c[B] * c[D];
    if (fabs(d) < MC_EPS) return f >= 0;
    return f * c[A] * d >= 0;
}

static int test_interior(const double *c, int mc_case, int config, int subconfig, int s)
{
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0, a, b;
    int edge = -1;
    if (mc_case == 4 || mc_case == 10) {
        a = (c[4] - c[0]) * (c[6] - c[2]) - (c[7] - c[3]) * (c[5] - c[1]);
        b = c[2] * (c[4] - c[0]) + c[0] * (c[6] - c[2]) - c[1] * (c[7] - c[3]) - c[3] * (c[5] - c[1]);
        t = -b / (2 * a + MC_EPS);           /* scikit-image adds its epsilon to the denominator */
        if (t < 0 || t > 1) return s > 0;
        At = c[0] + (c[4] - c[0]) * t;
        Bt = c[3] + (c[7] - c[3]) * t;
        Ct = c[2] + (c[6] - c[2]) * t;
        Dt = c[1] + (c[5] - c[1]) * t;
    } else {
        switch (mc_case) {
        case 6: edge = MC_TEST6[config][2]; break;
        case 7: edge = MC_TEST7[config][4]; break;
        case 12: edge = MC_TEST12[config][3]; break;
        case 13: edge = MC_TILING13_5_1[config][subconfig][0]; break;
        default: break;
        }
        /* e:(p,q,(b0,b1),(c0,c1),(d0,d1)) of App. B item 6 */
        static const int T[12][8] = {
            {0,1,3,2,7,6,4,5}, {1,2,0,3,4,7,5,6}, {2,3,1,0,5,4,6,7}, {3,0,2,1,6,5,7,4},
            {4,5,7,6,3,2,0,1}, {5,6,4,7,0,3,1,2}, {6,7,5,4,1,0,2,3}, {7,4,6,5,2,1,3,0},
            {0,4,3,7,2,6,1,5}, {1,5,0,4,3,7,2,6}, {2,6,1,5,0,4,3,7}, {3,7,2,6,1,5,0,4}};
        if (edge >= 0 && edge < 12) {
            const int *e = T[edge];
            t = c[e[0]] / (c[e[0]] - c[e[1]] + MC_EPS);
            At = 0;
            Bt = c[e[2]] + (c[e[3]] - c[e[2]]) * t;
            Ct = c[e[4]] + (c[e[5]] - c[e[4]]) * t;
            Dt = c[e[6]] + (c[e[7]] - c[e[6]]) * t;
        }
    }
    int test = 0;
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    case 7: case 11: case 13: case 14: case 15: return s < 0;
    case 5: if (At * Ct - Bt * Dt < MC_EPS) return s > 0; return 0;     /* skimage fall-through: False */
    case 10: if (At * Ct - Bt * Dt >= MC_EPS) return s > 0; return 0;
    }
    return 0;
}

/* Select the triangle row for a cell; returns the number of triangles. */
static int choose_tiling(const double *c, int index, const signed char **row)
{
    int mc_case = MC_CASES[index][0], cfg = MC_CASES[index][1], sc, k, s13;
    switch (mc_case) {
    case 1: *row = MC_TILING1[cfg]; return 1;
    case 2: *row = MC_TILING2[cfg]; return 2;
    case 3:
        if (test_face(c, MC_TEST3[cfg])) { *row = MC_TILING3_2[cfg]; return 4; }
        *row = MC_TILING3_1[cfg]; return 2;
    case 4:
        if (test_interior(c, 4, cfg, 0, MC_TEST4[cfg])) { *row = MC_TILING4_1[cfg]; return 2; }
        *row = MC_TILING4_2[cfg]; return 6;
    case 5: *row = MC_TILING5[cfg]; return 3;
    case 6:
        if (test_face(c, MC_TEST6[cfg][0])) { *row = MC_TILING6_2[cfg]; return 5; }
        if (test_interior(c, 6, cfg, 0, MC_TEST6[cfg][1])) { *row = MC_TILING6_1_1[cfg]; return 3; }
        *row = MC_TILING6_1_2[cfg]; return 9;
    case 7:
        sc = 0;
        for (k = 0; k < 3; ++k) if (test_face(c, MC_TEST7[cfg][k])) sc += 1 << k;
        switch (sc) {
        case 0: *row = MC_TILING7_1[cfg]; return 3;
        case 1: *row = MC_TILING7_2[cfg][0]; return 5;
        case 2: *row = MC_TILING7_2[cfg][1]; return 5;
        case 3: *row = MC_TILING7_3[cfg][0]; return 9;
        case 4: *row = MC_TILING7_2[cfg][2]; return 5;
        case 5: *row = MC_TILING7_3[cfg][1]; return 9;
        case 6: *row = MC_TILING7_3[cfg][2]; return 9;
        default:
            if (test_interior(c, 7, cfg, 0, MC_TEST7[cfg][3])) { *row = MC_TILING7_4_2[cfg]; return 9; }
            *row = MC_TILING7_4_1[cfg]; return 5;
        }
    case 8: *row = MC_TILING8[cfg]; return 2;
    case 9: *row = MC_TILING9[cfg]; return 4;
    case 10:
        if (test_face(c, MC_TEST10[cfg][0])) {
            if (test_face(c, MC_TEST10[cfg][1])) { *row = MC_TILING10_1_1_[cfg]; return 4; }
            *row = MC_TILING10_2[cfg]; return 8;
        }
        if (test_face(c, MC_TEST10[cfg][1])) { *row = MC_TILING10_2_[cfg]; return 8; }
        if (test_interior(c, 10, cfg, 0, MC_TEST10[cfg][2])) { *row = MC_TILING10_1_1[cfg]; return 4; }
        *row = MC_TILING10_1_2[cfg]; return 8;
    case 11: *row = MC_TILING11[cfg]; return 4;
    case 12:
        if (test_face(c, MC_TEST12[cfg][0])) {
            if (test_face(c, MC_TEST12[cfg][1])) { *row = MC_TILING12_1_1_[cfg]; return 4; }
            *row = MC_TILING12_2[cfg]; return 8;
        }
        if (test_face(c, MC_TEST12[cfg][1])) { *row = MC_TILING12_2_[cfg]; return 8; }
        if (test_interior(c, 12, cfg, 0, MC_TEST12[cfg][2])) { *row = MC_TILING12_1_1[cfg]; return 4; }
        *row = MC_TILING12_1_2[cfg]; return 8;
    case 13:
        sc = 0;
        for (k = 0; k < 6; ++k) if (test_face(c, MC_TEST13[cfg][k])) sc += 1 << k;
        s13 = MC_SUBCONFIG13[sc];
        if (s13 == 0) { *row = MC_TILING13_1[cfg]; return 4; }
        if (s13 <= 6) { *row = MC_TILING13_2[cfg][s13 - 1]; return 6; }
        if (s13 <= 18) { *row = MC_TILING13_3[cfg][s13 - 7]; return 10; }
        if (s13 <= 22) { *row = MC_TILING13_4[cfg][s13 - 19]; return 12; }
        if (s13 <= 26) {
            k = s13 - 23;
            if (test_interior(c, 13, cfg, k, MC_TEST13[cfg][6])) { *row = MC_TILING13_5_1[cfg][k]; return 6; }
            *row = MC_TILING13_5_2[cfg][k]; return 10;
        }
        if (s13 <= 38) { *row = MC_TILING13_3_[cfg][s13 - 27]; return 10; }
        if (s13 <= 44) { *row = MC_TILING13_2_[cfg][s13 - 39]; return 6; }
        *row = MC_TILING13_1_[cfg]; return 4;
    case 14: *row = MC_TILING14[cfg]; return 4;
    default: *row = 0; return 0;
    }
}

/* vol: float32 [n0][n1][n2] (array axes; internal z,y,x).  Two-call protocol: pass verts=faces=NULL to
 * count, then buffers with capacities.  verts: float32 [V][3] in array-axis order; faces: int32 [F][3]
 * with the 'descent' flip applied.  Returns 0, or -1 if a capacity is too small. */
int mc_lewiner_oracle(const float *vol, int n0, int n1, int n2, double level,
                      float *verts, int64_t vcap, int32_t *faces, int64_t fcap,
                      int64_t *nverts, int64_t *nfaces)
{
    const int nz = n0, ny = n1, nx = n2;
    const size_t layer = (size_t)ny * nx;
    /* vertex index tables: x- and y-edges of layers z (0) and z+1 (1); z-edges between them */
    int32_t *ex[2], *ey[2], *ez;
    ex[0] = malloc(layer * 4); ex[1] = malloc(layer * 4);
    ey[0] = malloc(layer * 4); ey[1] = malloc(layer * 4);
    ez = malloc(layer * 4);
    memset(ex[0], 0xff, layer * 4); memset(ey[0], 0xff, layer * 4);
    int64_t V = 0, Fc = 0;
    int rc = 0;
    for (int z = 0; z < nz - 1; ++z) {
        memset(ex[1], 0xff, layer * 4); memset(ey[1], 0xff, layer * 4); memset(ez, 0xff, layer * 4);
        for (int y = 0; y < ny - 1; ++y) {
            for (int x = 0; x < nx - 1; ++x) {
                double v[8], c[8];
                int index = 0;
                for (int i = 0; i < 8; ++i) {
                    v[i] = (double)vol[((size_t)(z + CORNER_OFF[i][2]) * ny + (y + CORNER_OFF[i][1])) * nx + (x + CORNER_OFF[i][0])];
                    c[i] = v[i] - level;
                    if (v[i] > level) index |= 1 << i;
                }
                if (index == 0 || index == 255) continue;
                const signed char *row = 0;
                int nt = choose_tiling(c, index, &row);
                int32_t centre = -1;
                for (int k = 0; k < 3 * nt; ++k) {
                    int e = row[k];
                    int32_t vid;
                    if (e == 12) {
                        if (centre < 0) {
                            double sw = 0, sx = 0, sy = 0, sz = 0;
                            for (int i = 0; i < 8; ++i) {
                                double w = 1.0 / (MC_EPS + fabs(c[i]));
                                sx += w * CORNER_OFF[i][0];
                                sy += w * CORNER_OFF[i][1];
                                sz += w * CORNER_OFF[i][2];
                                sw += w;
                            }
                            if (verts) {
                                if (V >= vcap) { rc = -1; goto done; }
                                verts[3 * V + 0] = (float)(z + sz / sw);
                                verts[3 * V + 1] = (float)(y + sy / sw);
                                verts[3 * V + 2] = (float)(x + sx / sw);
                            }
                            centre = (int32_t)V++;
                        }
                        vid = centre;
                    } else {
                        int a = EDGE_CORNERS[e][0], b = EDGE_CORNERS[e][1];
                        int ax = x + CORNER_OFF[a][0], ay = y + CORNER_OFF[a][1], az = z + CORNER_OFF[a][2];
                        int bx = x + CORNER_OFF[b][0], by = y + CORNER_OFF[b][1], bz = z + CORNER_OFF[b][2];
                        int lx = ax < bx ? ax : bx, ly = ay < by ? ay : by, lz = az < bz ? az : bz;
                        int32_t *slot;
                        int axis;
                        if (ax != bx) { axis = 0; slot = &ex[lz - z][(size_t)ly * nx + lx]; }
                        else if (ay != by) { axis = 1; slot = &ey[lz - z][(size_t)ly * nx + lx]; }
                        else { axis = 2; slot = &ez[(size_t)ly * nx + lx]; }
                        if (*slot < 0) {
                            /* lo = endpoint with the smaller coordinate along the edge axis */
                            int lo_is_a = (axis == 0 ? ax < bx : axis == 1 ? ay < by : az < bz);
                            /* strength-weighted interpolation, as scikit-image does it: w = 1/(eps+|v-level|);
                             * offset along the edge = w_hi / (w_lo + w_hi).  Equal to (level-v_lo)/(v_hi-v_lo)
                             * after float32 rounding except when an endpoint sits exactly on the level. */
                            double wlo = 1.0 / (MC_EPS + fabs(lo_is_a ? c[a] : c[b]));
                            double whi = 1.0 / (MC_EPS + fabs(lo_is_a ? c[b] : c[a]));
                            double t = whi / (wlo + whi);
                            if (verts) {
                                if (V >= vcap) { rc = -1; goto done; }
                                verts[3 * V + 0] = (float)(axis == 2 ? lz + t : (double)lz);
                                verts[3 * V + 1] = (float)(axis == 1 ? ly + t : (double)ly);
                                verts[3 * V + 2] = (float)(axis == 0 ? lx + t : (double)lx);
                            }
                            *slot = (int32_t)V++;
                        }
                        vid = *slot;
                    }
                    if (faces) {
                        if (Fc >= fcap) { rc = -1; goto done; }
                        /* 'descent': reverse the three indices of each triangle */
                        faces[3 * Fc + (2 - k % 3)] = vid;
                    }
                    if (k % 3 == 2) ++Fc;
                }
            }
        }
        int32_t *t0 = ex[0]; ex[0] = ex[1]; ex[1] = t0;
        t0 = ey[0]; ey[0] = ey[1]; ey[1] = t0;
    }
done:
    free(ex[0]); free(ex[1]); free(ey[0]); free(ey[1]); free(ez);
    *nverts = V; *nfaces = Fc;
    return rc;
}
