// The three pieces of shade() a path vertex is made of (MTPC/pathTracing.cpp:137-266), in the form the wavefront kernels use
// them: what the surface looks like at the hit, one light sample, and Russian roulette + nextRay.  Same arithmetic, operation
// for operation, as shade_path_from() (shade_path.hpp); tests compare the two pipelines bit for bit.
#pragma once
#include "dev_common.hpp"
#include "shade_common.hpp"

namespace mcpt {

// interpolated normal and diffuse colour at p on leaf `leaf` (:147-160, texture lookup Q9 / D7)
__device__ __forceinline__ void vertex_surface(const DScene& S, int leaf, const V3& p, const DMaterial* m, V3& pn, V3& kd)
{
    const DTri* tr = S.tris + leaf;
    const DTriShade* sh = S.shade + leaf;
    const V3 g = barycentric_s(ld3(tr->v1), ld3(tr->v2), ld3(tr->v3), p);
    pn = (ld3(sh->vn1) * g.x + ld3(sh->vn2) * g.y) + ld3(sh->vn3) * g.z;
    if (m->has_map) {
        const double row = sh->vt1[0] * g.x + sh->vt2[0] * g.y + sh->vt3[0] * g.z;
        const double col = sh->vt1[1] * g.x + sh->vt2[1] * g.y + sh->vt3[1] * g.z;
        const double irow = row - floor(row), icol = col - floor(col);
        int rr = (int)(irow * m->map_h), cc = (int)(icol * m->map_w);
        rr = rr < 0 ? 0 : (rr > m->map_h - 1 ? m->map_h - 1 : rr);
        cc = cc < 0 ? 0 : (cc > m->map_w - 1 ? m->map_w - 1 : cc);
        const uint8_t* px = S.texels + m->tex_offset + ((size_t)rr * m->map_w + cc) * 3;
        kd = mk((double)px[2] * MCPT_INV_255, (double)px[1] * MCPT_INV_255, (double)px[0] * MCPT_INV_255);
    } else kd = ld3(m->kd);
}

// Light l seen from the vertex (:166-232).  Returns the material the shadow ray must reach for the light to count, or -2 when
// the light is behind the surface (the only case in which the shadow ray's answer is not used); direction = the shadow ray's
// direction (origin p + 0.01 direction), c = the contribution if visible.  sample_mat carries over from light to light as in
// the reference (a light whose area sample fails keeps the previous light's material).
__device__ __forceinline__ int light_sample(const DScene& S, const RngKey& key, uint32_t depth, int l, const V3& p, const V3& pn, const V3& kd,
                                            int& sample_mat, V3& direction, V3& c)
{
    const DLight* lt = S.lights + l;
    V3 xl = mk(0, 0, 0), vn = mk(0, 0, 0);
    double u0, u1, u2, u3;
    uniform4(key, depth, (uint32_t)l, u0, u1, u2, u3);
    const double rnd = u0 * S.area0;                                            // frozen static u1 range (Q1)
    const int jt = pick_light_triangle(S.light_cdf + lt->first, lt->ntri, lt->cdf_sorted != 0, rnd);
    if (jt >= 0) {
        const DLightTri* q = S.light_tris + lt->first + jt;
        sample_mat = lt->material;
        const double isum = frcp(u1 + u2 + u3);
        const double p1 = u1 * isum, p2 = u2 * isum, p3 = u3 * isum;
        xl = (ld3(q->v1) * p1 + ld3(q->v2) * p2) + ld3(q->v3) * p3;
        vn = (ld3(q->vn1) * p1 + ld3(q->vn2) * p2) + ld3(q->vn3) * p3;
    }
    direction = normalized_s(xl - p);
    const double kd_dots = dot(direction, pn);
    if (!(kd_dots > 0)) return -2;
    // (the reference also divides by |direction|, a unit vector: 1 to within the two ulps this arithmetic is held to)
    const double cos_theta = fabs(dot(direction, vn) * frcp(norm_s(vn)));
    const double cos_theta_hat = fabs(kd_dots * frcp(norm_s(pn)));
    const double dd = norm_s(xl - p);
    const double dist = (1.0 < dd) ? dd : 1.0;                                  // std::max(1.0, distance)
    const V3 intensity = ((ld3(lt->radiance) * cos_theta) * cos_theta_hat) * (frcp(sqr(dist)) * lt->total_area);
    c = mk(kd.x * intensity.x * kd_dots * MCPT_INV_PI, kd.y * intensity.y * kd_dots * MCPT_INV_PI, kd.z * intensity.z * kd_dots * MCPT_INV_PI);
    return sample_mat;
}

#define MCPT_BT_NO_OFFSET 8         /* flag in a bounce type: the ray starts at the vertex itself (refraction, total reflection) */

// Russian roulette and nextRay (:3-11, :66-134, :234-263) at a vertex reached along -dir.  Returns -1 when the path ends here,
// else the ray type (| MCPT_BT_NO_OFFSET), with nd = direction of the bounce ray and wgt = kd / ks / 1.
__device__ __forceinline__ int bounce_sample(const RngKey& key, uint32_t depth, int nl, const DMaterial* m, const V3& dir, const V3& pn, const V3& kd,
                                             V3& nd, V3& wgt)
{
    if (depth + 1 >= MCPT_MAX_DEPTH_DEV) return -1;                            // D6
    double u_rr, u_fresnel, u_lobe, u_phi;
    uniform4(key, depth, (uint32_t)nl, u_rr, u_fresnel, u_lobe, u_phi);
    if (!(u_rr < MCPT_P_RR)) return -1;
    int btype = -1, at_vertex = 0;
    const V3 ks = ld3(m->ks);
    if (m->Ni > 1) {
        double n1, n2;
        const double cos_in = dot(neg(dir), pn);
        V3 normal;
        if (cos_in > 0) { normal = neg(pn); n1 = m->Ni; n2 = 1.0; }
        else { normal = pn; n1 = 1.0; n2 = m->Ni; }
        const double rf0 = sqr((n1 - n2) / (n1 + n2));
        const double fresnel = rf0 + (1.0f - rf0) * pow5(1.0f - fabs(cos_in));
        if (fresnel < u_fresnel) {
            V3 direction;
            at_vertex = MCPT_BT_NO_OFFSET;
            if (refract_dir(neg(dir), normal, n1 / n2, direction)) { nd = direction; btype = RT_TRANSMISSION; }
            else {
                const V3 incoming = neg(dir);
                nd = incoming - (normal * dot(incoming, normal)) * 2; btype = RT_SPECULAR;
            }
        }
    }
    if (btype < 0) {
        const double u_theta = uniform1(key, depth, (uint32_t)nl + 1u);
        const double ks_norm = norm_s(ks);
        if (ks_norm != 0 && norm_s(kd) * frcp(ks_norm) < u_lobe) {
            const V3 incoming = neg(dir);
            const V3 reflect = incoming - (pn * dot(incoming, pn)) * 2;
            nd = brdf_sample(u_phi, u_theta, reflect, RT_SPECULAR, m->Ns);
            btype = RT_SPECULAR;
        } else {
            nd = brdf_sample(u_phi, u_theta, pn, RT_DIFFUSE, m->Ns);
            btype = RT_DIFFUSE;
        }
    }
    wgt = btype == RT_DIFFUSE ? kd : (btype == RT_SPECULAR ? ks : mk(1, 1, 1));
    return btype | at_vertex;
}

}  // namespace mcpt
