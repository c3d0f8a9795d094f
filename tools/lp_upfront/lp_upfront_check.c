/* lp_upfront_check.c -- CPU check of the restructured ORCA linear programs (tools/lp_upfront/orca_lp_upfront.h) against
 * the oracle's sequential linearProgram2/3 (oracle/cagym_oracle.c: lp2 / lp3), BIT FOR BIT, on random and degenerate
 * half-plane sets.  Test infrastructure only (build: gcc -O2 -ffp-contract=off -o /tmp/lpc tools/lp_upfront/lp_upfront_check.c -lm).
 *
 * The restructuring: linearProgram1(i) depends only on line i, the lines before it and the
 * optimisation velocity - not on the running result - so R_i ("the optimum on line i", or "infeasible at i") is computed
 * for ALL lines up front as independent work; column c of a violation matrix says which lines i > c are violated by R_c
 * (column -1: by the start point); linearProgram2's sequential part is then a find-first-set walk over the columns.
 * linearProgram3 keeps its outer scan and solves each inner program the same way on the projected lines.
 */
#include "../oracle/cagym_oracle.c"
#include <stdio.h>
#include <string.h>

#define NMAX 40

static void clip_by(const orca_line* li, const orca_line* lj, float* tl, float* tr, int* bad) {
    const float den = detf(li->dx, li->dy, lj->dx, lj->dy);
    const float num = detf(lj->dx, lj->dy, li->px - lj->px, li->py - lj->py);
    if (fabsf(den) <= RVO_EPS) {
        if (num < 0.0f) *bad = 1;
    } else {
        const float t = num / den;
        if (den >= 0.0f) *tr = fminf(*tr, t);
        else *tl = fmaxf(*tl, t);
    }
}

/* R_i of the up-front formulation; have: bit k = line k exists (holes of the projected set) */
static int lp1_upfront(const orca_line* L, unsigned have, int i, float radius, float ox, float oy, int dir_opt, float* rx, float* ry) {
    const orca_line* li = &L[i];
    const float dot = li->px * li->dx + li->py * li->dy;
    const float disc = dot * dot + radius * radius - (li->px * li->px + li->py * li->py);
    if (disc < 0.0f) return 0;
    const float sq = sqrtf(disc);
    float tl = -dot - sq, tr = -dot + sq;
    int bad = 0;
    for (int k = 0; k < i; k++)
        if ((have >> k) & 1u) clip_by(li, &L[k], &tl, &tr, &bad);
    if (bad || tl > tr) return 0;
    float t;
    if (dir_opt) t = (ox * li->dx + oy * li->dy > 0.0f) ? tr : tl;
    else {
        t = li->dx * (ox - li->px) + li->dy * (oy - li->py);
        if (t < tl) t = tl;
        else if (t > tr) t = tr;
    }
    *rx = li->px + t * li->dx;
    *ry = li->py + t * li->dy;
    return 1;
}

/* linearProgram2 over the existing lines of L[0..n): returns the index of the failing line (n = success), result in rx, ry */
static int lp2_upfront(const orca_line* L, unsigned have, int n, float radius, float ox, float oy, int dir_opt, float* rx, float* ry) {
    float Rx[NMAX], Ry[NMAX], sx, sy;
    int feas[NMAX];
    unsigned col[NMAX + 1];
    if (dir_opt) { sx = ox * radius; sy = oy * radius; }
    else if (ox * ox + oy * oy > radius * radius) {
        float inv = 1.0f / sqrtf(ox * ox + oy * oy);
        sx = ox * inv * radius; sy = oy * inv * radius;
    } else { sx = ox; sy = oy; }
    for (int i = 0; i < n; i++) {
        Rx[i] = Ry[i] = 0.f;
        feas[i] = ((have >> i) & 1u) ? lp1_upfront(L, have, i, radius, ox, oy, dir_opt, &Rx[i], &Ry[i]) : 0;
    }
    /* column c + 1: lines i > c violated by X_c (X_-1 = the start point) */
    for (int c = -1; c < n; c++) {
        const float xx = c < 0 ? sx : Rx[c], yy = c < 0 ? sy : Ry[c];
        unsigned m = 0;
        for (int i = c + 1; i < n; i++)
            if (((have >> i) & 1u) && detf(L[i].dx, L[i].dy, L[i].px - xx, L[i].py - yy) > 0.0f) m |= 1u << i;
        col[c + 1] = m;
    }
    int c = -1;
    for (;;) {
        const unsigned m = col[c + 1];
        if (!m) break;
        const int i = __builtin_ctz(m);
        if (!feas[i]) { *rx = c < 0 ? sx : Rx[c]; *ry = c < 0 ? sy : Ry[c]; return i; }
        c = i;
    }
    *rx = c < 0 ? sx : Rx[c]; *ry = c < 0 ? sy : Ry[c];
    return n;
}

static void lp3_upfront(const orca_line* L, int n, int num_obst, int begin, float radius, float* rx, float* ry) {
    float distance = 0.0f;
    orca_line P[NMAX];
    for (int i = begin; i < n; i++) {
        if (detf(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry) > distance) {
            unsigned have = 0;
            for (int j = 0; j < num_obst; j++) { P[j] = L[j]; have |= 1u << j; }
            for (int j = num_obst; j < i; j++) {
                orca_line ln;
                float d = detf(L[i].dx, L[i].dy, L[j].dx, L[j].dy);
                if (fabsf(d) <= RVO_EPS) {
                    if (L[i].dx * L[j].dx + L[i].dy * L[j].dy > 0.0f) continue;  /* hole: no projected line j */
                    ln.px = 0.5f * (L[i].px + L[j].px);
                    ln.py = 0.5f * (L[i].py + L[j].py);
                } else {
                    float s = detf(L[j].dx, L[j].dy, L[i].px - L[j].px, L[i].py - L[j].py) / d;
                    ln.px = L[i].px + s * L[i].dx;
                    ln.py = L[i].py + s * L[i].dy;
                }
                float ddx = L[j].dx - L[i].dx, ddy = L[j].dy - L[i].dy;
                float inv = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
                ln.dx = ddx * inv; ln.dy = ddy * inv;
                P[j] = ln;
                have |= 1u << j;
            }
            const int np = i > num_obst ? i : num_obst;
            float qx, qy;
            if (lp2_upfront(P, have, np, radius, -L[i].dy, L[i].dx, 1, &qx, &qy) == np) { *rx = qx; *ry = qy; }
            distance = detf(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry);
        }
    }
}

static unsigned long long rs = 88172645463325252ull;
static double urand(void) {
    rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
    return (double)(rs >> 11) / 9007199254740992.0;
}

int main(void) {
    long n_lp3 = 0, n_fail = 0, n_cases = 0, n_holes = 0;
    for (int it = 0; it < 3000000; it++) {
        const int n = 1 + (int)(urand() * 19.99);
        orca_line L[NMAX];
        const int mode = (int)(urand() * 6);
        const float radius = (float)(0.3 + 1.2 * urand());
        for (int k = 0; k < n; k++) {
            double ang = 2 * M_PI * urand();
            if (mode == 1 && k > 0 && urand() < 0.4) {  /* parallel / anti-parallel / nearly parallel to an earlier line */
                int j = (int)(urand() * k);
                double e = urand() < 0.5 ? 0.0 : 2e-5 * (urand() - 0.5);
                ang = atan2((double)L[j].dy, (double)L[j].dx) + (urand() < 0.5 ? 0.0 : M_PI) + e;
                if (urand() < 0.3) { L[k] = L[j]; if (urand() < 0.5) { L[k].dx = -L[j].dx; L[k].dy = -L[j].dy; } continue; }
            }
            float dx = (float)cos(ang), dy = (float)sin(ang);
            float inv = 1.0f / sqrtf(dx * dx + dy * dy);
            L[k].dx = dx * inv; L[k].dy = dy * inv;
            double spread = mode == 2 ? 3.0 : (mode == 3 ? 0.2 : 1.2);  /* far points: infeasible discs; near: crowds */
            if (mode >= 4) {  /* crowd: half-planes that push away from the origin's surroundings -> linearProgram3 */
                double a2 = ang + M_PI / 2, off = 0.05 + 0.5 * urand();
                L[k].px = (float)(off * cos(a2)); L[k].py = (float)(off * sin(a2));
                if (mode == 5 && urand() < 0.5) { L[k].px = -L[k].px; L[k].py = -L[k].py; }
            } else {
                L[k].px = (float)(spread * (2 * urand() - 1)); L[k].py = (float)(spread * (2 * urand() - 1));
            }
        }
        float ox = (float)(1.6 * (2 * urand() - 1)), oy = (float)(1.6 * (2 * urand() - 1));
        const int num_obst = (mode == 5 && n > 2) ? (int)(urand() * 3) : 0;
        float ax, ay, bx, by;
        int fa = lp2(L, n, radius, ox, oy, 0, &ax, &ay);
        int fb = lp2_upfront(L, n >= 32 ? 0xffffffffu : ((1u << n) - 1u), n, radius, ox, oy, 0, &bx, &by);
        if (fa != fb || memcmp(&ax, &bx, 4) || memcmp(&ay, &by, 4)) {
            printf("LP2 MISMATCH it=%d n=%d fail %d vs %d  (%g %g) vs (%g %g)\n", it, n, fa, fb, ax, ay, bx, by);
            return 1;
        }
        if (fa < n) {
            n_fail++;
            lp3(L, n, num_obst, fa, radius, &ax, &ay);
            lp3_upfront(L, n, num_obst, fb, radius, &bx, &by);
            n_lp3++;
            if (memcmp(&ax, &bx, 4) || memcmp(&ay, &by, 4)) {
                printf("LP3 MISMATCH it=%d n=%d begin %d (%g %g) vs (%g %g)\n", it, n, fa, ax, ay, bx, by);
                return 1;
            }
        }
        n_cases++;
    }
    (void)n_holes;
    printf("ok: %ld cases, %ld infeasible linearProgram2 -> linearProgram3, all bit-identical\n", n_cases, n_lp3);
    return 0;
}
