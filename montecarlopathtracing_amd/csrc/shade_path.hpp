// shade() as a loop over the path's vertices, one lane per path (megakernel pipeline and the wavefront's finishing pass).
#pragma once
#include "dev_common.hpp"
#include "shade_common.hpp"
#include "trace_fast.hpp"
#include "kernels.hpp"

namespace mcpt {

// shade() (pathTracing.cpp:137-266) with the recursion unrolled into a loop: the recursion is a chain
// (one bounce per vertex), so L = sum_d T_d * Ldir_d with T_{d+1} = T_d * w_d / 0.6.
// Resumable: starts at vertex `depth0` with throughput T, gathered radiance L, arrival direction dir and arrival ray type
// in_type (depth0 = 0, T = 1, L = 0, in_type = TRANSMISSION for a fresh camera sample).  FAST selects the walk
// (trace_lane_fast with this lane's LDS stack, or the reference-shaped trace_closest); both give the same hits.
template <bool FAST>
__device__ void shade_path_from(const DScene& S, const RngKey& key, uint32_t depth0, V3 T, V3 L, V3 dir, int in_type, Hit hit, double out[3],
                                LaneStats& ls, int* lds_stack, int stride)
{
    const uint32_t nl = (uint32_t)S.num_lights;
    Work w = {0, 0};
    auto trace = [&](const Ray& r, Hit& h) -> bool {
        if constexpr (FAST) return trace_lane_fast(S, r, h, w, lds_stack, stride);
        else return trace_closest(S, r, h, w);
    };
    for (uint32_t depth = depth0;; depth++) {
        ls.shades++;
        if (depth > ls.depth) ls.depth = depth;
        const DTri* tr = S.tris + hit.leaf;
        const DMaterial* m = S.materials + tr->material;
        if (m->light >= 0) {                                                    // :141-144
            const V3 rad = ld3(S.lights[m->light].radiance);
            if (depth == 0) L = rad;
            else if (in_type != RT_DIFFUSE) L = L + mk(T.x * rad.x, T.y * rad.y, T.z * rad.z);   // :247-261
            break;
        }
        const V3 tv1 = ld3(tr->v1), tv2 = ld3(tr->v2), tv3 = ld3(tr->v3);
        const DTriShade* sh = S.shade + hit.leaf;
        const V3 g = barycentric_s(tv1, tv2, tv3, hit.p);
        const V3 pn = (ld3(sh->vn1) * g.x + ld3(sh->vn2) * g.y) + ld3(sh->vn3) * g.z;
        V3 kd;
        if (m->has_map) {                                                       // :147-160 (Q9)
            const double row = sh->vt1[0] * g.x + sh->vt2[0] * g.y + sh->vt3[0] * g.z;
            const double col = sh->vt1[1] * g.x + sh->vt2[1] * g.y + sh->vt3[1] * g.z;
            const double irow = row - floor(row), icol = col - floor(col);
            int rr = (int)(irow * m->map_h), cc = (int)(icol * m->map_w);
            rr = rr < 0 ? 0 : (rr > m->map_h - 1 ? m->map_h - 1 : rr);          // D7
            cc = cc < 0 ? 0 : (cc > m->map_w - 1 ? m->map_w - 1 : cc);
            const uint8_t* px = S.texels + m->tex_offset + ((size_t)rr * m->map_w + cc) * 3;
            kd = mk((double)px[2] * MCPT_INV_255, (double)px[1] * MCPT_INV_255, (double)px[0] * MCPT_INV_255);
        } else kd = ld3(m->kd);

        // direct illumination, :166-232
        V3 L_dir = mk(0, 0, 0);
        int sample_mat = -1;
        for (uint32_t i = 0; i < nl; i++) {
            const DLight* lt = S.lights + i;
            V3 xl = mk(0, 0, 0), vn = mk(0, 0, 0);
            double u0, u1, u2, u3;
            uniform4(key, depth, i, u0, u1, u2, u3);
            const double rnd = u0 * S.area0;                                    // frozen static u1 range (Q1)
            const int j = pick_light_triangle(S.light_cdf + lt->first, lt->ntri, lt->cdf_sorted != 0, rnd);
            if (j >= 0) {
                const DLightTri* q = S.light_tris + lt->first + j;
                sample_mat = lt->material;
                const double rnd1 = u1, rnd2 = u2, rnd3 = u3;
                const double isum = frcp(rnd1 + rnd2 + rnd3);
                const double p1 = rnd1 * isum, p2 = rnd2 * isum, p3 = rnd3 * isum;
                xl = (ld3(q->v1) * p1 + ld3(q->v2) * p2) + ld3(q->v3) * p3;
                vn = (ld3(q->vn1) * p1 + ld3(q->vn2) * p2) + ld3(q->vn3) * p3;
            }
            const V3 direction = normalized_s(xl - hit.p);
            double visibility = 1;
            Ray rl; rl.o = hit.p + direction * 0.01; rl.d = direction;
            Hit inter;
            const bool got = trace(rl, inter);
            ls.shadow++;
            const int inter_mat = got ? S.tris[inter.leaf].material : -1;
            if (inter_mat != sample_mat) visibility = 0;                        // :213
            if (dot(direction, pn) > 0) {
                const double cos_theta = fabs(dot(direction, vn) * frcp(norm_s(vn)));
                const double cos_theta_hat = fabs(dot(direction, pn) * frcp(norm_s(pn)));
                const double dd = norm_s(xl - hit.p);
                const double dist = (1.0 < dd) ? dd : 1.0;                      // std::max(1.0, distance)
                const V3 intensity = (((ld3(lt->radiance) * cos_theta) * cos_theta_hat) * (frcp(sqr(dist)) * lt->total_area)) * visibility;
                const double kd_dots = dot(direction, pn);
                if (kd_dots > 0) {
                    L_dir.x += kd.x * intensity.x * kd_dots * MCPT_INV_PI;
                    L_dir.y += kd.y * intensity.y * kd_dots * MCPT_INV_PI;
                    L_dir.z += kd.z * intensity.z * kd_dots * MCPT_INV_PI;
                }
            }
        }
        L = L + mk(T.x * L_dir.x, T.y * L_dir.y, T.z * L_dir.z);

        // indirect illumination, :234-263
        if (depth + 1 >= MCPT_MAX_DEPTH_DEV) break;                             // D6
        double u_rr, u_fresnel, u_lobe, u_phi;
        uniform4(key, depth, nl, u_rr, u_fresnel, u_lobe, u_phi);               // slots 4nl (RR), 4nl+1 (FRESNEL), 4nl+2 (LOBE), 4nl+3 (PHI)
        if (!(u_rr < MCPT_P_RR)) break;                                         // russian_Roulette :3-11
        // nextRay, :66-134
        Ray nr; int type = -1;
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
                if (refract_dir(neg(dir), normal, n1 / n2, direction)) { nr.o = hit.p; nr.d = direction; type = RT_TRANSMISSION; }
                else {
                    const V3 incoming = neg(dir);
                    nr.o = hit.p; nr.d = incoming - (normal * dot(incoming, normal)) * 2; type = RT_SPECULAR;
                }
            }
        }
        if (type < 0) {
            const double u_theta = uniform1(key, depth, nl + 1u);               // slot 4nl+4 (THETA)
            const double ks_norm = norm_s(ks);
            V3 direction;
            if (ks_norm != 0 && norm_s(kd) * frcp(ks_norm) < u_lobe) {
                const V3 incoming = neg(dir);
                const V3 reflect = incoming - (pn * dot(incoming, pn)) * 2;
                direction = brdf_sample(u_phi, u_theta, reflect, RT_SPECULAR, m->Ns);
                type = RT_SPECULAR;
            } else {
                direction = brdf_sample(u_phi, u_theta, pn, RT_DIFFUSE, m->Ns);
                type = RT_DIFFUSE;
            }
            nr.o = hit.p + direction * 0.01; nr.d = direction;
        }
        Hit next;
        ls.bounce++;
        if (!trace(nr, next)) break;
        const V3 wgt = type == RT_DIFFUSE ? kd : (type == RT_SPECULAR ? ks : mk(1, 1, 1));
        T = mk(T.x * wgt.x * MCPT_INV_P_RR, T.y * wgt.y * MCPT_INV_P_RR, T.z * wgt.z * MCPT_INV_P_RR);
        hit = next; dir = neg(nr.d); in_type = type;
    }
    ls.nodes += w.nodes; ls.tris += w.tris;
    out[0] = L.x; out[1] = L.y; out[2] = L.z;
}

// a fresh camera sample through the reference-shaped walk (megakernel pipeline)
__device__ __forceinline__ void shade_path(const DScene& S, const RngKey& key, V3 view_dir, Hit hit, double out[3], LaneStats& ls)
{
    shade_path_from<false>(S, key, 0u, mk(1, 1, 1), mk(0, 0, 0), neg(view_dir), RT_TRANSMISSION, hit, out, ls, nullptr, 0);
}

}  // namespace mcpt
