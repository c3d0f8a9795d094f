/* cagym_oracle_grid.c -- TEST INFRASTRUCTURE (CPU oracle), twin of k_occupancy_grid (csrc/cagym_kernels.h).
 * OccupancyGridSensor.sense (gym_collision_avoidance/envs/sensors/OccupancyGridSensor.py:70-98, 131-143) with
 * Map.world_coordinates_to_map_indices (Map.py:40-47) and Map.getSubmapByIndices (Map.py:81-105): the 300x300
 * occupancy raster as float is rotated about the agent's cell by -heading with cv2.getRotationMatrix2D +
 * cv2.warpAffine (bilinear, constant-0 border), the 60x60 window around the agent is cut out and cast to bool.
 *
 * PARITY UNPINNED: OpenCV (cv2) is not installed here and the reference holds no stored output of this sensor.
 * warpAffine is restated from OpenCV 4.x imgproc (imgwarp.cpp, WarpAffineInvoker / remapBilinear): the matrix is
 * inverted in double, source coordinates are fixed point with AB_BITS = 10 and INTER_BITS = 5 (1/32 pixel,
 * round_delta = 16, cvRound = round-half-even), and a destination pixel is the weighted sum of the 2x2 source
 * patch with weights (1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx fy, fx = (X & 31)/32.  After astype(bool) only "which
 * patch pixels have a positive weight" matters. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MAPD 300
#define SUB 60
#define AB_BITS 10
#define INTER_BITS 5
#define CAO_PI 3.14159265358979323846 /* np.pi == CV_PI */

static int cv_round(double v) { return (int)lrint(v); } /* default rounding mode: half to even, as cvRound */

static int src_at(const uint8_t* map, int x, int y) { /* BORDER_CONSTANT, value 0 */
    if (x < 0 || y < 0 || x >= MAPD || y >= MAPD) return 0;
    return map[y * MAPD + x] != 0;
}

static void submap_start(int c, int* start) { /* Map.getSubmapByIndices, one axis (span 60, map 300) */
    double s = (double)c - floor(SUB / 2.0);
    long long si = (long long)s; /* int(): truncation toward zero */
    if (si < 0) si = 0;
    long long e = si + SUB;
    if (e > MAPD - 1) { e = MAPD - 1; si = e - SUB; }
    *start = (int)si;
}

/* static_map [300,300] row-major (row = gx, col = gy of Map.py); out [60,60] */
void cao_occupancy_grid(const uint8_t* static_map, double px, double py, double heading, uint8_t* out) {
    const double cell = 0.1, origin = (30 / 2.) / cell;
    double fgx = floor(origin - py / cell), fgy = floor(origin + px / cell);
    fgx = fgx < -1e6 ? -1e6 : (fgx > 1e6 ? 1e6 : fgx);
    fgy = fgy < -1e6 ? -1e6 : (fgy > 1e6 ? 1e6 : fgy);
    const int gx = (int)fgx, gy = (int)fgy;
    int sx0, sy0;
    submap_start(gx, &sx0); /* rows */
    submap_start(gy, &sy0); /* cols */
    /* cv2.getRotationMatrix2D(center=(gy, gx), angle=-heading*180/pi, scale=1) */
    double angle = -heading * 180 / CAO_PI;
    angle *= CAO_PI / 180;
    const double alpha = cos(angle), beta = sin(angle), cx = (double)gy, cy = (double)gx;
    double M[6] = {alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy};
    /* warpAffine: dst(x, y) = src(M^-1 (x, y, 1)) */
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    const int AB_SCALE = 1 << AB_BITS, round_delta = AB_SCALE / (1 << INTER_BITS) / 2;
    for (int r = 0; r < SUB; r++) {
        const int y = sx0 + r;
        const int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        const int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int c = 0; c < SUB; c++) {
            const int x = sy0 + c;
            const int X = (X0 + cv_round(M[0] * x * AB_SCALE)) >> (AB_BITS - INTER_BITS);
            const int Y = (Y0 + cv_round(M[3] * x * AB_SCALE)) >> (AB_BITS - INTER_BITS);
            const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS, fx = X & 31, fy = Y & 31;
            int v = src_at(static_map, sx, sy);
            if (fx) v |= src_at(static_map, sx + 1, sy);
            if (fy) v |= src_at(static_map, sx, sy + 1);
            if (fx && fy) v |= src_at(static_map, sx + 1, sy + 1);
            out[r * SUB + c] = (uint8_t)v;
        }
    }
}
