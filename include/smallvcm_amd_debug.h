/*
 * smallvcm_amd_debug.h -- parity / test entry points of libsmallvcm_amd.so.
 *
 * NOT part of the drop-in boundary (that is include/smallvcm_amd.h): nothing a renderer host needs is here.
 * These symbols let tests/ read the intermediate state the reference keeps in VertexCM::mLightVertices and
 * HashGrid (src/vertexcm.hxx:1021-1028, src/hashgrid.hxx:205-214), evaluate the numeric specification
 * (detmath, Philox) on the device and on the host, and check the POD sizes the ctypes
 * mirror assumes.  tests/test_abi.py requires every exported vcm_* symbol to be declared in one of the two
 * headers, and every declared symbol to be exported.
 */
#ifndef SMALLVCM_AMD_DEBUG_H
#define SMALLVCM_AMD_DEBUG_H

#include "smallvcm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Hash grid of the last iteration (HashGrid::mCellEnds / mIndices, hashgrid.hxx:205-214):
 * cellStart (nCells+1 ints, cellStart[c+1] == mCellEnds[c]), sortedIndex (grid position -> record index =
 * mIndices, nRecords ints), bbox (mBBoxMin, mBBoxMax: 6 floats).  Any pointer may be NULL. */
int vcm_debug_read_grid(vcm_ctx *ctx, int *cellStart, int *sortedIndex, float *bbox6, long long *nRecords);

/* The local merge records of the last iteration (VCM_MERGE_RECORD_FLOATS floats each, reference vertex order),
 * host copy; `count` from vcm_light_records. */
int vcm_debug_read_records(vcm_ctx *ctx, float *out, long long count);

/* Element-wise evaluation ON THE DEVICE of the numeric specification: op 0 sinf(a), 1 cosf(a), 2 powf(a,b)
 * (detmath.h), 3 a/b, 4 sqrtf(a) (correctly rounded), 5 a*b+a as separate mul and add (no contraction). */
int vcm_debug_numeric_spec(int op, int n, const float *a, const float *b, float *out);
/* nFloats consecutive floats of nPaths paths of the counter-based stream (philox.h), on the device */
int vcm_debug_philox_spec(unsigned seed, unsigned iter, unsigned kind, int nPaths, int nFloats, float *out);

/* The same definitions evaluated on the host (the radius schedule uses powf on the host, vertexcm.hxx:296) */
float vcm_host_sinf(float x);
float vcm_host_cosf(float x);
float vcm_host_powf(float x, float y);
float vcm_host_path_float(unsigned seed, unsigned iter, unsigned path, unsigned kind, unsigned k);

/* Function-level known answers (T0): record i = VCM_KAT_FLOATS input floats -> VCM_KAT_FLOATS output floats of ONE
 * call of a device function, evaluated on the device with the context's scene (smallvcm_amd/csrc/vcm_kat.h has the
 * field layout per op; unused fields are 0).  oracle/ref_driver.cpp answers the same records with the reference's
 * classes (ref_kat), tests/host_emul with the device functions compiled for the host. */
#define VCM_KAT_FLOATS 16
enum {
    VCM_KAT_INTERSECT = 0,        /* in: org, dir, tmin -> hit, dist, matID, lightID, normal                 scene.hxx:53-70 */
    VCM_KAT_OCCLUDED = 1,         /* in: point, dir, tmax -> occluded                                        scene.hxx:72-85 */
    VCM_KAT_BSDF_EVAL = 2,        /* in: rayDir, normal, matID, dirGen -> valid, isDelta, contProb, f, cosGen, dirPdf,
                                     revPdf, Pdf(dir), Pdf(rev), WorldDirFix, CosThetaFix                    bsdf.hxx:95-180 */
    VCM_KAT_BSDF_SAMPLE = 3,      /* in: rayDir, normal, matID, rnd3, fixIsLight -> valid, f, dirGen, pdfW, cosGen, event
                                                                                                             bsdf.hxx:191-257 */
    VCM_KAT_LIGHT_EMIT = 4,       /* in: light, dirRnd2, posRnd2 -> energy, position, direction, emissionPdfW, directPdfA,
                                     cosLight, IsFinite, IsDelta                                             lights.hxx */
    VCM_KAT_LIGHT_ILLUMINATE = 5, /* in: light, receiver, rnd2 -> radiance, dirToLight, distance, directPdfW, emissionPdfW,
                                     cosAtLight */
    VCM_KAT_LIGHT_RADIANCE = 6,   /* in: light, rayDir, hitPoint -> radiance, directPdfA, emissionPdfW */
    VCM_KAT_CAMERA = 7,           /* in: raster x, y, world point -> ray dir, raster of the point, CheckRaster
                                                                                                             camera.hxx:95-117 */
    VCM_KAT_OPS = 8
};
int vcm_debug_kat(vcm_ctx *ctx, int op, int n, const float *in, float *out);

/* sizeof the PODs of smallvcm_amd.h as the library was compiled */
unsigned vcm_sizeof_scene_desc(void);
unsigned vcm_sizeof_stats(void);

#ifdef __cplusplus
}
#endif
#endif /* SMALLVCM_AMD_DEBUG_H */
