// The 16 real spherical-harmonics basis values of a unit direction (utils/sh_utils.py:57-112 of the reference:
// colour = sum_k basis_k(dir) * sh[k]), zero above the active degree.  Same expressions as the SH backward of
// preprocess_bwd.hip, which also needs their derivatives; used by the factored SH optimiser step (adam.hip).
#pragma once
#include "gsr_constants.h"

__device__ __forceinline__ void sh_basis16(int deg, float x, float y, float z, float* basis) {
#pragma unroll
    for (int k = 0; k < 16; ++k) basis[k] = 0.f;
    basis[0] = GSR_SH_C0;
    if (deg > 0) {
        basis[1] = -GSR_SH_C1 * y; basis[2] = GSR_SH_C1 * z; basis[3] = -GSR_SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            basis[4] = GSR_SH_C2_0 * xy; basis[5] = GSR_SH_C2_1 * yz;
            basis[6] = GSR_SH_C2_2 * (2.f * zz - xx - yy);
            basis[7] = GSR_SH_C2_3 * xz; basis[8] = GSR_SH_C2_4 * (xx - yy);
            if (deg > 2) {
                basis[9] = GSR_SH_C3_0 * y * (3.f * xx - yy);
                basis[10] = GSR_SH_C3_1 * xy * z;
                basis[11] = GSR_SH_C3_2 * y * (4.f * zz - xx - yy);
                basis[12] = GSR_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                basis[13] = GSR_SH_C3_4 * x * (4.f * zz - xx - yy);
                basis[14] = GSR_SH_C3_5 * z * (xx - yy);
                basis[15] = GSR_SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

// The same with the derivatives of every basis function w.r.t. the (unit) direction components, treated as independent
// (the caller chains through the normalisation): what d(rgb)/d(dir) is built from.
__device__ __forceinline__ void sh_basis16_grad(int deg, float x, float y, float z, float* basis, float* dbx, float* dby,
                                                float* dbz) {
    sh_basis16(deg, x, y, z, basis);
#pragma unroll
    for (int k = 0; k < 16; ++k) { dbx[k] = 0.f; dby[k] = 0.f; dbz[k] = 0.f; }
    if (deg > 0) {
        dby[1] = -GSR_SH_C1; dbz[2] = GSR_SH_C1; dbx[3] = -GSR_SH_C1;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dbx[4] = GSR_SH_C2_0 * y; dby[4] = GSR_SH_C2_0 * x;
            dby[5] = GSR_SH_C2_1 * z; dbz[5] = GSR_SH_C2_1 * y;
            dbx[6] = GSR_SH_C2_2 * -2.f * x; dby[6] = GSR_SH_C2_2 * -2.f * y; dbz[6] = GSR_SH_C2_2 * 4.f * z;
            dbx[7] = GSR_SH_C2_3 * z; dbz[7] = GSR_SH_C2_3 * x;
            dbx[8] = GSR_SH_C2_4 * 2.f * x; dby[8] = GSR_SH_C2_4 * -2.f * y;
            if (deg > 2) {
                dbx[9] = GSR_SH_C3_0 * 6.f * xy; dby[9] = GSR_SH_C3_0 * (3.f * xx - 3.f * yy);
                dbx[10] = GSR_SH_C3_1 * yz; dby[10] = GSR_SH_C3_1 * xz; dbz[10] = GSR_SH_C3_1 * xy;
                dbx[11] = GSR_SH_C3_2 * -2.f * xy; dby[11] = GSR_SH_C3_2 * (4.f * zz - xx - 3.f * yy); dbz[11] = GSR_SH_C3_2 * 8.f * yz;
                dbx[12] = GSR_SH_C3_3 * -6.f * xz; dby[12] = GSR_SH_C3_3 * -6.f * yz; dbz[12] = GSR_SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
                dbx[13] = GSR_SH_C3_4 * (4.f * zz - 3.f * xx - yy); dby[13] = GSR_SH_C3_4 * -2.f * xy; dbz[13] = GSR_SH_C3_4 * 8.f * xz;
                dbx[14] = GSR_SH_C3_5 * 2.f * xz; dby[14] = GSR_SH_C3_5 * -2.f * yz; dbz[14] = GSR_SH_C3_5 * (xx - yy);
                dbx[15] = GSR_SH_C3_6 * (3.f * xx - 3.f * yy); dby[15] = GSR_SH_C3_6 * -6.f * xy;
            }
        }
    }
}
