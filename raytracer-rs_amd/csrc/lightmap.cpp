// lightmap.cpp — conservative depth cube maps around the point lights (see lightmap.hpp).
#include "lightmap.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace mi355rt {
namespace {

struct V3 { double x, y, z; };
inline V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 add(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 mul(V3 a, double s) { return { a.x * s, a.y * s, a.z * s }; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline double len(V3 a) { return std::sqrt(dot(a, a)); }
inline double comp(V3 a, int k) { return k == 0 ? a.x : k == 1 ? a.y : a.z; }

// distance from the origin to triangle abc (closest-point regions, Ericson, "Real-Time Collision Detection" 5.1.5)
double origin_triangle_distance(V3 a, V3 b, V3 c)
{
    const V3 ab = sub(b, a), ac = sub(c, a), ap = mul(a, -1.0);
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) return len(a);
    const V3 bp = mul(b, -1.0);
    const double d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.0 && d4 <= d3) return len(b);
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) return len(add(a, mul(ab, d1 / (d1 - d3))));
    const V3 cp = mul(c, -1.0);
    const double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) return len(c);
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) return len(add(a, mul(ac, d2 / (d2 - d6))));
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) return len(add(b, mul(sub(c, b), (d4 - d3) / ((d4 - d3) + (d5 - d6)))));
    const double den = va + vb + vc;
    if (!(std::fabs(den) > 0.0)) return -1.0;                      // degenerate triangle: the caller falls back to a cruder lower bound
    const double v = vb / den, w = vc / den;
    return len(add(a, add(mul(ab, v), mul(ac, w))));
}

// Sutherland-Hodgman against one half space f(p) >= 0, f linear
template <class F> int clip(const V3* in, int n, V3* out, F f)
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const V3 A = in[i], B = in[(i + 1) % n];
        const double fa = f(A), fb = f(B);
        if (fa >= 0.0) out[m++] = A;
        if ((fa >= 0.0) != (fb >= 0.0)) { const double t = fa / (fa - fb); out[m++] = add(A, mul(sub(B, A), t)); }
    }
    return m;
}

inline float round_down(double v)          // largest float <= v (v >= 0)
{
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, 0.0f);
    return f;
}

}  // namespace

// face f = 2 * major axis + (direction component negative); the two other axes (a, b) in ascending order
static const int kAxisA[3] = { 1, 0, 0 }, kAxisB[3] = { 2, 2, 1 };

void build_light_map(const float* tri_verts, uint32_t ntri, const float light[3], double pad, uint32_t res, LightMap& out)
{
    const uint32_t R = res;
    out.res = R;
    out.dist2.assign((size_t)6 * R * R, std::numeric_limits<float>::infinity());
    out.nearest = std::numeric_limits<double>::infinity();
    const V3 L = { light[0], light[1], light[2] };
    const double rho = std::sqrt(2.0) / (double)R * 1.001;          // angular radius of a texel (gnomonic coordinates shrink angles, never stretch them)
    const double cos_rho = std::cos(rho), sin_rho = std::sin(rho);
    for (uint32_t t = 0; t < ntri; ++t) {
        const float* v = tri_verts + 9 * (size_t)t;
        const V3 P[3] = { sub({ v[0], v[1], v[2] }, L), sub({ v[3], v[4], v[5] }, L), sub({ v[6], v[7], v[8] }, L) };
        double dmin = origin_triangle_distance(P[0], P[1], P[2]);
        {   // the closest point is never farther than the nearest vertex, nor closer than it by more than the triangle's extent: a degenerate
            // triangle (or a NaN) gets that cruder LOWER bound
            const double size = std::max(len(sub(P[1], P[0])), std::max(len(sub(P[2], P[0])), len(sub(P[2], P[1]))));
            const double near_vertex = std::min(len(P[0]), std::min(len(P[1]), len(P[2])));
            const double lower = std::max(near_vertex - size, 0.0);
            if (!(dmin >= 0.0) || !std::isfinite(dmin)) dmin = lower;
            dmin = std::min(std::max(dmin, lower), near_vertex);
            if (!(dmin == dmin)) dmin = 0.0;
        }
        out.nearest = std::min(out.nearest, dmin);
        // the supporting plane: n . x = c; h = its distance from the light; nh = unit normal pointing from the light towards the plane
        const V3 n = cross(sub(P[1], P[0]), sub(P[2], P[0]));
        const double nn = len(n);
        const bool has_plane = nn > 0.0 && std::isfinite(nn);
        double h = 0.0; V3 nh = { 0, 0, 0 };
        if (has_plane) { const double c = dot(n, P[0]) / nn; h = std::fabs(c); nh = mul(n, (c < 0.0 ? -1.0 : 1.0) / nn); }
        for (int face = 0; face < 6; ++face) {
            const int m = face >> 1, a = kAxisA[m], b = kAxisB[m];
            const double sg = (face & 1) ? -1.0 : 1.0;
            // clip the triangle to the face's frustum (slightly widened), project, bound
            V3 poly[2][16]; int np = 3;
            poly[0][0] = P[0]; poly[0][1] = P[1]; poly[0][2] = P[2];
            const double eps = 1e-12 * (1.0 + len(P[0])), dl = 1e-6;
            int cur = 0;
            np = clip(poly[cur], np, poly[cur ^ 1], [&](V3 q) { return sg * comp(q, m) - eps; }); cur ^= 1; if (np == 0) continue;
            np = clip(poly[cur], np, poly[cur ^ 1], [&](V3 q) { return sg * comp(q, m) * (1.0 + dl) - comp(q, a); }); cur ^= 1; if (np == 0) continue;
            np = clip(poly[cur], np, poly[cur ^ 1], [&](V3 q) { return sg * comp(q, m) * (1.0 + dl) + comp(q, a); }); cur ^= 1; if (np == 0) continue;
            np = clip(poly[cur], np, poly[cur ^ 1], [&](V3 q) { return sg * comp(q, m) * (1.0 + dl) - comp(q, b); }); cur ^= 1; if (np == 0) continue;
            np = clip(poly[cur], np, poly[cur ^ 1], [&](V3 q) { return sg * comp(q, m) * (1.0 + dl) + comp(q, b); }); cur ^= 1; if (np == 0) continue;
            double u0 = 1e300, u1 = -1e300, v0 = 1e300, v1 = -1e300, mmin = 1e300;
            for (int k = 0; k < np; ++k) {
                const double mm = sg * comp(poly[cur][k], m);
                if (!(mm > 0.0)) { mmin = 0.0; continue; }
                mmin = std::min(mmin, mm);
                const double uu = comp(poly[cur][k], a) / mm, vv = comp(poly[cur][k], b) / mm;
                u0 = std::min(u0, uu); u1 = std::max(u1, uu); v0 = std::min(v0, vv); v1 = std::max(v1, vv);
            }
            long i0 = 0, i1 = (long)R - 1, j0 = 0, j1 = (long)R - 1;
            if (mmin > 0.0 && u0 <= u1) {
                // a world-space shift of `pad` moves u = a / m by at most pad * (1 + |u|) / m <= 2.5 * pad / m inside the (widened) frustum
                const double mu = 2.5 * pad / mmin + 1e-7;
                i0 = (long)std::floor((std::max(u0 - mu, -1.0) * 0.5 + 0.5) * R - 0.01); i1 = (long)std::floor((std::min(u1 + mu, 1.0) * 0.5 + 0.5) * R + 0.01);
                j0 = (long)std::floor((std::max(v0 - mu, -1.0) * 0.5 + 0.5) * R - 0.01); j1 = (long)std::floor((std::min(v1 + mu, 1.0) * 0.5 + 0.5) * R + 0.01);
                i0 = std::max(i0, 0L); j0 = std::max(j0, 0L); i1 = std::min(i1, (long)R - 1); j1 = std::min(j1, (long)R - 1);
            }                                                       // else: the triangle reaches the light's own plane: the whole face
            float* fm = out.dist2.data() + (size_t)face * R * R;
            for (long i = i0; i <= i1; ++i) {
                const double uc = ((double)i + 0.5) / R * 2.0 - 1.0;
                for (long j = j0; j <= j1; ++j) {
                    double lb = dmin;
                    if (has_plane) {
                        // every direction of the texel makes an angle >= theta_c - rho with nh, so the plane (and with it the triangle) is at
                        // least h / cos(theta_c - rho) away in all of them
                        const double vc = ((double)j + 0.5) / R * 2.0 - 1.0;
                        double w[3]; w[m] = sg; w[a] = uc; w[b] = vc;
                        const double wl = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                        const double cos_c = (w[0] * nh.x + w[1] * nh.y + w[2] * nh.z) / wl;
                        if (cos_c > -0.999999) {
                            const double sin_c = std::sqrt(std::max(0.0, 1.0 - cos_c * cos_c));
                            const double cos_lo = cos_c >= cos_rho ? 1.0 : cos_c * cos_rho + sin_c * sin_rho;      // cos(max(theta_c - rho, 0))
                            if (cos_lo > 1e-4) lb = std::max(lb, h / cos_lo);
                            else lb = std::max(lb, h * 1e4);
                        }
                    }
                    const double d = std::max(lb - pad, 0.0);
                    const float val = round_down(d * d * (1.0 - 1e-6));
                    float& dst = fm[(size_t)i * R + (size_t)j];
                    if (val < dst) dst = val;
                }
            }
        }
    }
}

double point_triangle_distance_lower(const double p[3], const float* v)
{
    const V3 L = { p[0], p[1], p[2] };
    const V3 P[3] = { sub({ v[0], v[1], v[2] }, L), sub({ v[3], v[4], v[5] }, L), sub({ v[6], v[7], v[8] }, L) };
    double dmin = origin_triangle_distance(P[0], P[1], P[2]);
    const double size = std::max(len(sub(P[1], P[0])), std::max(len(sub(P[2], P[0])), len(sub(P[2], P[1]))));
    const double near_vertex = std::min(len(P[0]), std::min(len(P[1]), len(P[2])));
    const double lower = std::max(near_vertex - size, 0.0);
    if (!(dmin >= 0.0) || !std::isfinite(dmin)) dmin = lower;
    dmin = std::min(std::max(dmin, lower), near_vertex);
    return dmin == dmin ? dmin : 0.0;
}

uint64_t light_map_work(const float* tri_verts, uint32_t ntri, const float light[3], uint32_t res)
{
    // texel updates build_light_map would make at resolution `res`: the solid angle of every triangle's bounding cone, in texels (6 R^2 texels = 4 pi)
    const V3 L = { light[0], light[1], light[2] };
    double total = 0.0;
    for (uint32_t t = 0; t < ntri; ++t) {
        const float* v = tri_verts + 9 * (size_t)t;
        const V3 P[3] = { sub({ v[0], v[1], v[2] }, L), sub({ v[3], v[4], v[5] }, L), sub({ v[6], v[7], v[8] }, L) };
        const V3 c = mul(add(P[0], add(P[1], P[2])), 1.0 / 3.0);
        const double r = std::max(len(sub(P[0], c)), std::max(len(sub(P[1], c)), len(sub(P[2], c)))), d = len(c);
        double frac = 1.0;                                          // of the sphere
        if (d > r * 1.0001) { const double s = r / d; frac = std::min(1.0, s * s * 0.5 + 4.0 / ((double)res * res)); }   // bounding rectangles over-cover: x2
        total += frac * 6.0 * (double)res * res;
    }
    return (uint64_t)std::min(total, 1e18);
}

}  // namespace mi355rt
