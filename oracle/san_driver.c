/* oracle/san_driver.c -- TEST INFRASTRUCTURE.  Runs the C restatement (riou_oracle.c, linked in) under AddressSanitizer and
 * UndefinedBehaviorSanitizer on a box file written by tests/test_oracle_sanitizers.py (the reference's
 * device function writes up to 24 points into 8-element buffers -- rotate_polygon_nms_kernel.cu:163-201 -- which is undefined
 * behaviour there; the restatement uses 24-point buffers and must be clean on the same inputs, including degenerate boxes).
 *   san_driver <boxes.bin> <n> <thr> <out.bin>
 * boxes.bin: n x 6 float32 (cx, cy, w, h, angle, score).  out.bin: int64 K, K x int64 keep list (lazy NMS), int64 K2, K2 x int64
 * (bit-matrix NMS, n <= 4096), then the n x n IoU matrix of the first min(n, 256) boxes as float32. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void oracle_riou_matrix(const float *b1, int n1, int stride1, const float *b2, int n2, int stride2, float *out);
int oracle_rnms(const float *dets, int n, int stride, float thr, int64_t *keep_out, int nthreads, int64_t *pairs_out);
int oracle_rnms_bitmatrix(const float *dets, int n, int stride, float thr, int64_t *keep_out, uint64_t *mask_out);

int main(int argc, char **argv) {
    if (argc != 5) return 2;
    const int n = atoi(argv[2]);
    const float thr = (float)atof(argv[3]);
    float *d = (float *)malloc(sizeof(float) * 6 * (size_t)n);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(d, sizeof(float) * 6, (size_t)n, f) != (size_t)n) return 3;
    fclose(f);
    int64_t *keep = (int64_t *)malloc(sizeof(int64_t) * (size_t)n), *keep2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t pairs = 0;
    const int64_t k = oracle_rnms(d, n, 6, thr, keep, 1, &pairs);
    const int64_t k2 = n <= 4096 ? oracle_rnms_bitmatrix(d, n, 6, thr, keep2, NULL) : -1;
    const int m = n < 256 ? n : 256;
    float *iou = (float *)malloc(sizeof(float) * (size_t)m * m);
    oracle_riou_matrix(d, m, 6, d, m, 6, iou);
    f = fopen(argv[4], "wb");
    if (!f) return 4;
    fwrite(&k, 8, 1, f);
    fwrite(keep, 8, (size_t)k, f);
    fwrite(&k2, 8, 1, f);
    if (k2 > 0) fwrite(keep2, 8, (size_t)k2, f);
    fwrite(iou, 4, (size_t)m * m, f);
    fclose(f);
    free(d); free(keep); free(keep2); free(iou);
    return 0;
}
