/*
 * mcpt_oracle.c -- CPU restatement (plain C, fp64) of the reference hot path.
 * TEST INFRASTRUCTURE ONLY: see mcpt_oracle.h for who may use it, the pinning status
 * ("PARITY UNPINNED" for traversal/shading bitwise; statistically pinned against the
 * reference's published renders) and the list of documented deviations D1..D8.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * Citations are reference paths relative to /root/reference.
 */
#include "mcpt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ vector type
 * MTPC/sceneManagement.h:18-86 (class Vertex).  Operation order is the reference's. */
typedef struct { double x, y, z; } vec3;

static inline vec3 v3(double x, double y, double z) { vec3 r = { x, y, z }; return r; }
static inline vec3 vadd(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 vsub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 vmul(vec3 a, double t) { return v3(a.x * t, a.y * t, a.z * t); }
static inline vec3 vdiv(vec3 a, double m) { return v3(a.x / m, a.y / m, a.z / m); }
static inline vec3 vneg(vec3 a) { return v3(-a.x, -a.y, -a.z); }
static inline double vdot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* sceneManagement.h:68-74 */
static inline vec3 vcross(vec3 a, vec3 b)
{
    return v3(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y);
}
static inline double vnorm(vec3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline vec3 vnormalize(vec3 a) { double d = vnorm(a); return v3(a.x / d, a.y / d, a.z / d); }

/* sceneManagement.cpp:3-15 */
static inline double dmin3(double p1, double p2, double p3)
{
    if (p1 <= p2 && p1 <= p3) return p1;
    else if (p2 <= p1 && p2 <= p3) return p2;
    else return p3;
}
static inline double dmax3(double p1, double p2, double p3)
{
    if (p1 >= p2 && p1 >= p3) return p1;
    else if (p2 >= p1 && p2 >= p3) return p2;
    else return p3;
}

/* ------------------------------------------------------------------ scene types */
typedef struct {
    vec3 v1, v2, v3;
    vec3 vn1, vn2, vn3;
    double vt1[2], vt2[2], vt3[2];
    vec3 norm;
    int material;            /* index into materials (reference: std::string name) */
    uint32_t morton;
    int orig;                /* index in .obj order (the reference Face carries none) */
} Face;

typedef struct {
    char name[64];
    vec3 kd, ks;
    double Ns, Ni;
    int has_map, map_w, map_h;
    uint8_t* bgr;            /* OpenCV layout: rows x cols x (B,G,R) */
    int nf, capf;
    int* faces;              /* .obj-order indices of this material's faces (Material::f) */
    int light;               /* index of the light carrying this name, or -1 */
} Material;

typedef struct {
    char name[64];
    vec3 radiance;
    int material;
    double total_area;
    double* cdf;             /* running area sum per triangle (pathTracing.cpp:177-184) */
} Light;

typedef struct { double b[6]; int level; int leaf; } Node;  /* b = max_x,max_y,max_z,min_x,min_y,min_z */

struct orc_scene {
    int nv, nvn, nvt, nf, nm, nl;
    vec3 *v, *vn; double (*vt)[2];
    Face* f;                 /* .obj order */
    int* order;              /* leaf k -> .obj face index (stable Morton sort, D2) */
    Material* m; int capm;
    Light* l; int capl;
    vec3 eye, look_at, up; double fovy; int width, height;
    /* BVH (BVH.cpp:44-85) */
    orc_bvh_info bi;
    Node* nodes;
    double area0;            /* total area of light 0: the frozen range of the static u1 (Q1) */
    int walk_mode;           /* ORC_TRACE_* the integrator's rays are walked with (same hits; ALIAS = the reference's visit counts) */
};

/* ------------------------------------------------------------------ RNG seam (D1) */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo)
{
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32); *lo = (uint32_t)p;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* One uniform on the open interval (0,1) with 32 random bits: word slot & 3 of the Philox block with counter
 * (pixel, sample, depth<<16 | slot>>2, 'MCPT') and key = seed, as (word + 0.5) * 2^-32 (exact in a double). */
double orc_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t slot)
{
    uint32_t ctr[4] = { pixel, sample, (depth << 16) | (slot >> 2), 0x4D435054u };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    return ((double)o[slot & 3u] + 0.5) * (1.0 / 4294967296.0);
}

/* ------------------------------------------------------------------ Morton key
 * MTPC/morton code.cpp:3-32, MTPC/morton code.h:6-7 (MINP -1, MAXP 4). */
static uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline float fmax_std(float a, float b) { return (a < b) ? b : a; }   /* std::max */
static inline float fmin_std(float a, float b) { return (b < a) ? b : a; }   /* std::min */

uint32_t orc_morton_code(float x, float y, float z)
{
    /* (x - MINP) / (MAXP - MINP): int operands promoted to float */
    float mx = (x - (float)(-1)) / (float)(4 - (-1));
    float my = (y - (float)(-1)) / (float)(4 - (-1));
    float mz = (z - (float)(-1)) / (float)(4 - (-1));
    mx = fmin_std(fmax_std(mx * 1024.0f, 0.0f), 1023.0f);
    my = fmin_std(fmax_std(my * 1024.0f, 0.0f), 1023.0f);
    mz = fmin_std(fmax_std(mz * 1024.0f, 0.0f), 1023.0f);
    uint32_t xx = expand_bits((uint32_t)mx);
    uint32_t yy = expand_bits((uint32_t)my);
    uint32_t zz = expand_bits((uint32_t)mz);
    return xx * 4 + yy * 2 + zz;
}

/* ------------------------------------------------------------------ loader helpers
 * The reference parses with std::string::substr/find + atof/atoi.  sub() mimics
 * substr(pos,len) on a C string (len<0 == npos); find_ch mimics find() returning -1 (npos
 * truncated to int, as the reference stores it in an int). */
static int find_ch(const char* s, char c)
{
    const char* p = strchr(s, c);
    return p ? (int)(p - s) : -1;
}
static double atof_n(const char* s, int len)   /* atof(s.substr(0,len)) */
{
    char buf[128];
    size_t n = strlen(s);
    if (len >= 0 && (size_t)len < n) n = (size_t)len;
    if (n > sizeof(buf) - 1) n = sizeof(buf) - 1;
    memcpy(buf, s, n); buf[n] = 0;
    return atof(buf);
}
static int atoi_n(const char* s, int len)
{
    char buf[64];
    size_t n = strlen(s);
    if (len >= 0 && (size_t)len < n) n = (size_t)len;
    if (n > sizeof(buf) - 1) n = sizeof(buf) - 1;
    memcpy(buf, s, n); buf[n] = 0;
    return atoi(buf);
}
/* readline = readline.substr(blank+1) with blank == -1 -> whole string */
static const char* after(const char* s, int pos) { return s + (pos + 1); }

/* "a b c" -> three atof's exactly like sceneManagement.cpp:27-36 */
static vec3 parse3(const char* s)
{
    vec3 r; int blank;
    blank = find_ch(s, ' '); r.x = atof_n(s, blank); s = after(s, blank);
    blank = find_ch(s, ' '); r.y = atof_n(s, blank); s = after(s, blank);
    r.z = atof(s);
    return r;
}

static int starts(const char* line, const char* key)  /* strcmp(line.substr(0,n), key)==0 */
{
    return strncmp(line, key, strlen(key)) == 0;
}

static char* read_line(FILE* fp, char* buf, int cap)
{
    if (!fgets(buf, cap, fp)) return NULL;
    size_t n = strlen(buf);
    while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;   /* D4 */
    return buf;
}

static int find_material(const orc_scene* s, const char* name)
{
    for (int i = 0; i < s->nm; i++) if (strcmp(s->m[i].name, name) == 0) return i;
    return -1;
}

static int load_ppm(const char* path, int* w, int* h, uint8_t** bgr)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return -1;
    char magic[3] = { 0 }; int maxv = 0;
    if (fscanf(fp, "%2s", magic) != 1 || strcmp(magic, "P6") != 0) { fclose(fp); return -1; }
    int vals[3], got = 0;
    while (got < 3) {
        int c = fgetc(fp);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(fp); continue; }
        if (c == EOF) { fclose(fp); return -1; }
        if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
        ungetc(c, fp);
        if (fscanf(fp, "%d", &vals[got]) != 1) { fclose(fp); return -1; }
        got++;
    }
    fgetc(fp);
    *w = vals[0]; *h = vals[1]; maxv = vals[2];
    if (maxv != 255) { fclose(fp); return -1; }
    size_t n = (size_t)(*w) * (size_t)(*h) * 3;
    uint8_t* rgb = (uint8_t*)malloc(n);
    if (fread(rgb, 1, n, fp) != n) { free(rgb); fclose(fp); return -1; }
    fclose(fp);
    for (size_t i = 0; i < n; i += 3) { uint8_t t = rgb[i]; rgb[i] = rgb[i + 2]; rgb[i + 2] = t; } /* -> BGR */
    *bgr = rgb;
    return 0;
}

/* MTPC/sceneManagement.cpp:17-74 */
static int read_mtl(orc_scene* s, const char* fn, const char* texdir, char* err, int errlen)
{
    FILE* fp = fopen(fn, "r");
    if (!fp) { snprintf(err, errlen, "cannot open %s", fn); return -1; }
    char line[4096];
    Material* pm = NULL;
    while (read_line(fp, line, sizeof line)) {
        if (starts(line, "newmtl")) {
            if (s->nm == s->capm) { s->capm = s->capm ? 2 * s->capm : 16; s->m = (Material*)realloc(s->m, sizeof(Material) * s->capm); }
            const char* name = strlen(line) >= 7 ? line + 7 : "";
            int ex = find_material(s, name);          /* map[] overwrite: keep one entry per name */
            pm = &s->m[ex >= 0 ? ex : s->nm];
            memset(pm, 0, sizeof *pm);
            snprintf(pm->name, sizeof pm->name, "%s", name);
            pm->Ns = 1; pm->Ni = 1;                   /* D8 */
            pm->light = -1;
            if (ex < 0) s->nm++;
        } else if (starts(line, "Kd")) {
            if (!pm) goto nomat;
            pm->kd = parse3(strlen(line) >= 3 ? line + 3 : "");
        } else if (starts(line, "Ks")) {
            if (!pm) goto nomat;
            pm->ks = parse3(strlen(line) >= 3 ? line + 3 : "");
        } else if (starts(line, "Ns")) {
            if (!pm) goto nomat;
            pm->Ns = atof(strlen(line) >= 3 ? line + 3 : "");
        } else if (starts(line, "Ni")) {
            if (!pm) goto nomat;
            pm->Ni = atof(strlen(line) >= 3 ? line + 3 : "");
        } else if (starts(line, "map_Kd")) {
            if (!pm) goto nomat;
            const char* tex = strlen(line) >= 7 ? line + 7 : "";
            char path[4096];
            snprintf(path, sizeof path, "%s/%s.ppm", texdir ? texdir : ".", tex);
            pm->has_map = 1;
            if (load_ppm(path, &pm->map_w, &pm->map_h, &pm->bgr) != 0) {
                snprintf(err, errlen, "cannot read decoded texture %s", path);
                fclose(fp); return -1;
            }
        }
    }
    fclose(fp);
    return 0;
nomat:
    fclose(fp);
    snprintf(err, errlen, "%s: property before newmtl", fn);
    return -1;
}

/* MTPC/sceneManagement.cpp:76-189 */
static int read_obj(orc_scene* s, const char* fn, char* err, int errlen)
{
    FILE* fp = fopen(fn, "r");
    if (!fp) { snprintf(err, errlen, "cannot open %s", fn); return -1; }
    char linebuf[4096];
    int capv = 0, capvn = 0, capvt = 0, capf = 0;
    int material = -1;
    int have_material = 0;
    while (read_line(fp, linebuf, sizeof linebuf)) {
        const char* line = linebuf;
        if (line[0] == 'v' && line[1] == ' ') {
            if (s->nv == capv) { capv = capv ? 2 * capv : 1024; s->v = (vec3*)realloc(s->v, sizeof(vec3) * capv); }
            s->v[s->nv++] = parse3(line + 2);
        } else if (line[0] == 'v' && line[1] == 'n' && line[2] == ' ') {
            if (s->nvn == capvn) { capvn = capvn ? 2 * capvn : 1024; s->vn = (vec3*)realloc(s->vn, sizeof(vec3) * capvn); }
            s->vn[s->nvn++] = parse3(line + 3);
        } else if (line[0] == 'v' && line[1] == 't' && line[2] == ' ') {
            if (s->nvt == capvt) { capvt = capvt ? 2 * capvt : 1024; s->vt = (double(*)[2])realloc(s->vt, sizeof(double[2]) * capvt); }
            const char* q = line + 3; int blank;
            blank = find_ch(q, ' '); double a = atof_n(q, blank); q = after(q, blank);
            blank = find_ch(q, ' '); double b = atof_n(q, blank);
            s->vt[s->nvt][0] = a; s->vt[s->nvt][1] = b; s->nvt++;
        } else if (starts(line, "usemtl")) {
            const char* name = strlen(line) >= 7 ? line + 7 : "";
            material = find_material(s, name);
            have_material = 1;
            if (material < 0) { snprintf(err, errlen, "usemtl '%s' not in .mtl", name); fclose(fp); return -1; }
        } else if (line[0] == 'f' && line[1] == ' ') {
            if (!have_material || material < 0) { snprintf(err, errlen, "face before usemtl"); fclose(fp); return -1; }
            int idx[9]; int slash = 0, blank;
            const char* q = line + 2;
            for (int c = 0; c < 3; c++) {
                slash = find_ch(q, '/'); idx[c * 3 + 0] = atoi_n(q, slash) - 1; q = after(q, slash);
                slash = find_ch(q, '/'); idx[c * 3 + 1] = atoi_n(q, slash) - 1; q = after(q, slash);
                if (c < 2) { blank = find_ch(q, ' '); idx[c * 3 + 2] = atoi_n(q, blank) - 1; q = after(q, blank); }
                else { idx[c * 3 + 2] = atoi_n(q, slash) - 1; }   /* sceneManagement.cpp:165: substr(0, slash) with the STALE slash */
            }
            for (int c = 0; c < 3; c++) {
                if (idx[c * 3] < 0 || idx[c * 3] >= s->nv || idx[c * 3 + 1] < 0 || idx[c * 3 + 1] >= s->nvn ||
                    idx[c * 3 + 2] < 0 || idx[c * 3 + 2] >= s->nvt) {
                    snprintf(err, errlen, "face %d: index out of range", s->nf); fclose(fp); return -1;
                }
            }
            if (s->nf == capf) { capf = capf ? 2 * capf : 4096; s->f = (Face*)realloc(s->f, sizeof(Face) * capf); }
            Face* f = &s->f[s->nf];
            /* 2nd index -> vn[], 3rd -> vt[] (sceneManagement.cpp:136-165) */
            f->v1 = s->v[idx[0]]; f->vn1 = s->vn[idx[1]]; f->vt1[0] = s->vt[idx[2]][0]; f->vt1[1] = s->vt[idx[2]][1];
            f->v2 = s->v[idx[3]]; f->vn2 = s->vn[idx[4]]; f->vt2[0] = s->vt[idx[5]][0]; f->vt2[1] = s->vt[idx[5]][1];
            f->v3 = s->v[idx[6]]; f->vn3 = s->vn[idx[7]]; f->vt3[0] = s->vt[idx[8]][0]; f->vt3[1] = s->vt[idx[8]][1];
            f->material = material;
            f->orig = s->nf;
            /* calNorm: sceneManagement.cpp:408-412 */
            f->norm = vnormalize(vcross(vsub(f->v1, f->v2), vsub(f->v3, f->v1)));
            /* center + Morton: sceneManagement.cpp:176-179 (double -> float at the call) */
            vec3 center = vdiv(vadd(vadd(f->v1, f->v2), f->v3), 3);
            f->morton = orc_morton_code((float)center.x, (float)center.y, (float)center.z);
            Material* pm = &s->m[material];
            if (pm->nf == pm->capf) { pm->capf = pm->capf ? 2 * pm->capf : 64; pm->faces = (int*)realloc(pm->faces, sizeof(int) * pm->capf); }
            pm->faces[pm->nf++] = s->nf;
            s->nf++;
        }
    }
    fclose(fp);
    return 0;
}

/* MTPC/sceneManagement.cpp:191-262 */
static int read_camera(orc_scene* s, const char* fn, char* err, int errlen)
{
    FILE* fp = fopen(fn, "r");
    if (!fp) { snprintf(err, errlen, "cannot open %s", fn); return -1; }
    char line[4096];
    while (read_line(fp, line, sizeof line)) {
        size_t n = strlen(line);
        if (starts(line, "eye")) s->eye = parse3(n >= 4 ? line + 4 : "");
        else if (starts(line, "lookat")) s->look_at = parse3(n >= 7 ? line + 7 : "");
        else if (starts(line, "up")) s->up = parse3(n >= 3 ? line + 3 : "");
        else if (starts(line, "fovy")) s->fovy = atof(n >= 5 ? line + 5 : "");
        else if (starts(line, "width")) s->width = atoi(n >= 6 ? line + 6 : "");
        else if (starts(line, "height")) s->height = atoi(n >= 7 ? line + 7 : "");
        else if (starts(line, "mtlname")) {
            const char* q = n >= 8 ? line + 8 : "";
            if (s->nl == s->capl) { s->capl = s->capl ? 2 * s->capl : 8; s->l = (Light*)realloc(s->l, sizeof(Light) * s->capl); }
            Light* L = &s->l[s->nl];
            memset(L, 0, sizeof *L);
            int blank = find_ch(q, ' ');
            size_t ln = blank < 0 ? strlen(q) : (size_t)blank;
            if (ln > sizeof(L->name) - 1) ln = sizeof(L->name) - 1;
            memcpy(L->name, q, ln); L->name[ln] = 0;
            q = after(q, blank);
            L->radiance = parse3(q);
            s->nl++;
        }
    }
    fclose(fp);
    return 0;
}

/* Face::calAera, MTPC/sceneManagement.cpp:399-406 */
static double face_area(const Face* f)
{
    double a = vnorm(vsub(f->v2, f->v1)), b = vnorm(vsub(f->v3, f->v1)), c = vnorm(vsub(f->v3, f->v2));
    double cos_c = (a * a + b * b - c * c) / (2 * a * b);
    double sin_c = sqrt(1 - pow(cos_c, 2));
    double aera = a * b * sin_c / 2;
    return aera;
}

static int count_set_bits(int x) { int r = 0; for (int i = 0; i < 32; i++) { if (x & 1) r++; x >>= 1; } return r; } /* BVH.cpp:7-15 */

/* BVH::findIndex, MTPC/BVH.cpp:99-104 */
static inline int find_index(const orc_bvh_info* b, int i, int l)
{
    int Lvl = b->Lv >> (b->Level - l + 1);
    int Nvl = 2 * Lvl - count_set_bits(Lvl);
    return i - Nvl;
}
int orc_find_index(const orc_scene* s, int i, int l) { return find_index(&s->bi, i, l); }

/* BVH::haveRightSubtree, MTPC/BVH.cpp:126-132 */
static inline int have_right(const orc_bvh_info* b, int node, int l)
{
    long lim = (1L << (l + 2)) - 1 - (b->Lv >> (b->Level - l - 1));
    return !(2L * node + 2 >= lim);
}

static void leaf_box(Node* n, const Face* f)   /* BVHNode::findBondingBox(Face&), BVH.cpp:87-97 */
{
    n->b[0] = dmax3(f->v1.x, f->v2.x, f->v3.x);
    n->b[1] = dmax3(f->v1.y, f->v2.y, f->v3.y);
    n->b[2] = dmax3(f->v1.z, f->v2.z, f->v3.z);
    n->b[3] = dmin3(f->v1.x, f->v2.x, f->v3.x);
    n->b[4] = dmin3(f->v1.y, f->v2.y, f->v3.y);
    n->b[5] = dmin3(f->v1.z, f->v2.z, f->v3.z);
}
static inline double std_max(double a, double b) { return (a < b) ? b : a; }
static inline double std_min(double a, double b) { return (b < a) ? b : a; }

/* MTPC/BVH.cpp:44-85 */
static int build_bvh(orc_scene* s, char* err, int errlen)
{
    orc_bvh_info* b = &s->bi;
    int t = s->nf;
    if (t <= 0) { snprintf(err, errlen, "scene has no faces"); return -1; }
    int Lc = 1; while (Lc < t) Lc <<= 1;          /* pow(2, ceil(log2(t))) */
    b->t = t; b->Lc = Lc; b->Lv = Lc - t; b->Nc = 2 * Lc - 1;
    b->Nv = 2 * b->Lv - count_set_bits(b->Lv);
    b->Nr = 2 * t - 1 + count_set_bits(b->Lv);
    int Level = 0; while ((1 << (Level + 1)) <= b->Nc) Level++;   /* floor(log2(Nc)) */
    b->Level = Level;
    s->nodes = (Node*)calloc((size_t)b->Nr, sizeof(Node));
    for (int l = Level; l >= 0; l--) {
        int current_level_v = b->Lv >> (Level - l);
        int start = (1 << l) - 1, end = (1 << (l + 1)) - 1 - current_level_v;
        int k = 0;
        for (int i = start; i < end; i++) {
            Node* n = &s->nodes[find_index(b, i, l)];
            n->level = l;
            if (l == Level) {
                n->leaf = k;
                leaf_box(n, &s->f[s->order[k]]);
                k++;
            } else {
                n->leaf = -1;
                const Node* c1 = &s->nodes[find_index(b, 2 * i + 1, l + 1)];
                if (have_right(b, i, l)) {
                    const Node* c2 = &s->nodes[find_index(b, 2 * i + 2, l + 1)];
                    for (int a = 0; a < 3; a++) n->b[a] = std_max(c1->b[a], c2->b[a]);
                    for (int a = 3; a < 6; a++) n->b[a] = std_min(c1->b[a], c2->b[a]);
                } else {
                    for (int a = 0; a < 6; a++) n->b[a] = c1->b[a];
                }
            }
        }
    }
    return 0;
}

orc_scene* orc_scene_load(const char* prefix, const char* texture_dir, char* err, int errlen)
{
    char fn[4096];
    char ebuf[256];
    if (!err) { err = ebuf; errlen = sizeof ebuf; }
    err[0] = 0;
    orc_scene* s = (orc_scene*)calloc(1, sizeof *s);
    /* read_scene order: mtl, obj, camera (sceneManagement.cpp:264-274) */
    snprintf(fn, sizeof fn, "%s.mtl", prefix);
    if (read_mtl(s, fn, texture_dir, err, errlen)) goto fail;
    snprintf(fn, sizeof fn, "%s.obj", prefix);
    if (read_obj(s, fn, err, errlen)) goto fail;
    snprintf(fn, sizeof fn, "%s.camera", prefix);
    if (read_camera(s, fn, err, errlen)) goto fail;
    if (s->nf == 0) { snprintf(err, errlen, "scene has no faces"); goto fail; }
    /* lights -> materials */
    for (int i = 0; i < s->nl; i++) {
        int m = find_material(s, s->l[i].name);
        if (m < 0) { snprintf(err, errlen, "light '%s' is not a material", s->l[i].name); goto fail; }
        s->l[i].material = m;
        s->m[m].light = i;      /* light_map[name] = l : the last one wins */
        Material* pm = &s->m[m];
        s->l[i].cdf = (double*)malloc(sizeof(double) * (pm->nf ? pm->nf : 1));
        double total = 0;
        for (int j = 0; j < pm->nf; j++) { total += face_area(&s->f[pm->faces[j]]); s->l[i].cdf[j] = total; }
        s->l[i].total_area = total;
    }
    s->area0 = s->nl ? s->l[0].total_area : 0.0;
    /* MTPC.cpp:44 sort by morton (stable, D2) */
    s->order = (int*)malloc(sizeof(int) * s->nf);
    {
        /* stable merge sort on (key, index) */
        int n = s->nf;
        int* a = s->order; int* tmp = (int*)malloc(sizeof(int) * n);
        for (int i = 0; i < n; i++) a[i] = i;
        for (int w = 1; w < n; w *= 2) {
            for (int lo = 0; lo < n; lo += 2 * w) {
                int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
                int i = lo, j = mid, k = lo;
                while (i < mid && j < hi) {
                    if (s->f[a[j]].morton < s->f[a[i]].morton) tmp[k++] = a[j++]; else tmp[k++] = a[i++];
                }
                while (i < mid) tmp[k++] = a[i++];
                while (j < hi) tmp[k++] = a[j++];
            }
            memcpy(a, tmp, sizeof(int) * n);
        }
        free(tmp);
    }
    if (build_bvh(s, err, errlen)) goto fail;
    return s;
fail:
    orc_scene_free(s);
    return NULL;
}

void orc_scene_free(orc_scene* s)
{
    if (!s) return;
    for (int i = 0; i < s->nm; i++) { free(s->m[i].faces); free(s->m[i].bgr); }
    for (int i = 0; i < s->nl; i++) free(s->l[i].cdf);
    free(s->v); free(s->vn); free(s->vt); free(s->f); free(s->order); free(s->m); free(s->l); free(s->nodes);
    free(s);
}

void orc_scene_set_resolution(orc_scene* s, int w, int h) { s->width = w; s->height = h; }
void orc_scene_set_walk_mode(orc_scene* s, int mode) { s->walk_mode = mode; }

/* Rebuilds the BVH over another order of the leaves that share a Morton key (D2): ORC_ORDER_STABLE = .obj order (the
 * oracle's default), ORC_ORDER_LIBSTDCXX = what std::sort (MTPC/MTPC.cpp:44) leaves in a g++/libstdc++ build. */
extern void orc_std_sort_order(const uint32_t* keys, int n, int32_t* order);
int orc_scene_set_leaf_order(orc_scene* s, int which)
{
    char err[128];
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)s->nf);
    for (int i = 0; i < s->nf; i++) keys[i] = s->f[i].morton;
    if (which == ORC_ORDER_LIBSTDCXX) orc_std_sort_order(keys, s->nf, s->order);
    else {                                                   /* stable: (key, index) ascending */
        int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)s->nf);
        orc_std_sort_order(keys, s->nf, tmp);                /* any key-sorted permutation, then equal keys by index */
        int i = 0;
        while (i < s->nf) {
            int j = i;
            while (j < s->nf && keys[tmp[j]] == keys[tmp[i]]) j++;
            for (int a = i + 1; a < j; a++) { int32_t v = tmp[a]; int b = a - 1; while (b >= i && tmp[b] > v) { tmp[b + 1] = tmp[b]; b--; } tmp[b + 1] = v; }
            i = j;
        }
        for (int k = 0; k < s->nf; k++) s->order[k] = tmp[k];
        free(tmp);
    }
    free(keys);
    for (int k = 1; k < s->nf; k++) if (s->f[s->order[k - 1]].morton > s->f[s->order[k]].morton) return -1;
    free(s->nodes); s->nodes = NULL;
    return build_bvh(s, err, sizeof err);
}
int orc_num_faces(const orc_scene* s) { return s->nf; }
int orc_num_materials(const orc_scene* s) { return s->nm; }
int orc_num_lights(const orc_scene* s) { return s->nl; }

void orc_get_camera(const orc_scene* s, double cam[10], int wh[2])
{
    cam[0] = s->eye.x; cam[1] = s->eye.y; cam[2] = s->eye.z;
    cam[3] = s->look_at.x; cam[4] = s->look_at.y; cam[5] = s->look_at.z;
    cam[6] = s->up.x; cam[7] = s->up.y; cam[8] = s->up.z; cam[9] = s->fovy;
    wh[0] = s->width; wh[1] = s->height;
}

void orc_get_faces(const orc_scene* s, double* g, int32_t* material, uint32_t* morton)
{
    for (int i = 0; i < s->nf; i++) {
        const Face* f = &s->f[i];
        double* o = g + (size_t)i * 27;
        const vec3* vs[6] = { &f->v1, &f->v2, &f->v3, &f->vn1, &f->vn2, &f->vn3 };
        for (int k = 0; k < 6; k++) { o[k * 3] = vs[k]->x; o[k * 3 + 1] = vs[k]->y; o[k * 3 + 2] = vs[k]->z; }
        o[18] = f->vt1[0]; o[19] = f->vt1[1]; o[20] = f->vt2[0]; o[21] = f->vt2[1]; o[22] = f->vt3[0]; o[23] = f->vt3[1];
        o[24] = f->norm.x; o[25] = f->norm.y; o[26] = f->norm.z;
        if (material) material[i] = f->material;
        if (morton) morton[i] = f->morton;
    }
}
void orc_get_leaf_order(const orc_scene* s, int32_t* o) { for (int i = 0; i < s->nf; i++) o[i] = s->order[i]; }
void orc_get_bvh_info(const orc_scene* s, orc_bvh_info* b) { *b = s->bi; }
void orc_get_bvh_nodes(const orc_scene* s, double* box6, int32_t* level, int32_t* leaf_face)
{
    for (int i = 0; i < s->bi.Nr; i++) {
        for (int a = 0; a < 6; a++) box6[(size_t)i * 6 + a] = s->nodes[i].b[a];
        if (level) level[i] = s->nodes[i].level;
        if (leaf_face) leaf_face[i] = s->nodes[i].leaf >= 0 ? s->order[s->nodes[i].leaf] : -1;
    }
}
void orc_get_material(const orc_scene* s, int m, char name[64], double r[8], int32_t fl[4])
{
    const Material* pm = &s->m[m];
    memcpy(name, pm->name, 64);
    r[0] = pm->kd.x; r[1] = pm->kd.y; r[2] = pm->kd.z; r[3] = pm->ks.x; r[4] = pm->ks.y; r[5] = pm->ks.z;
    r[6] = pm->Ns; r[7] = pm->Ni;
    fl[0] = pm->has_map; fl[1] = pm->map_w; fl[2] = pm->map_h; fl[3] = pm->light;
}
void orc_get_light(const orc_scene* s, int i, char name[64], double rad[3], int32_t* material, double* area)
{
    memcpy(name, s->l[i].name, 64);
    rad[0] = s->l[i].radiance.x; rad[1] = s->l[i].radiance.y; rad[2] = s->l[i].radiance.z;
    *material = s->l[i].material; *area = s->l[i].total_area;
}

/* ------------------------------------------------------------------ hit tests */
typedef struct { vec3 o, d; } RayT;

/* intersect(Ray&, boundingBox&), MTPC/sceneManagement.cpp:340-391 */
static int box_hit(const RayT* r, const double* b)
{
    double txmin, txmax, tymin, tymax, tzmin, tzmax;
    txmin = (b[3] - r->o.x) / r->d.x;
    txmax = (b[0] - r->o.x) / r->d.x;
    tymin = (b[4] - r->o.y) / r->d.y;
    tymax = (b[1] - r->o.y) / r->d.y;
    tzmin = (b[5] - r->o.z) / r->d.z;
    tzmax = (b[2] - r->o.z) / r->d.z;
    if (txmin > txmax) { double tmp = txmin; txmin = txmax; txmax = tmp; }
    if (tymin > tymax) { double tmp = tymin; tymin = tymax; tymax = tmp; }
    if (tzmin > tzmax) { double tmp = tzmin; tzmin = tzmax; tzmax = tmp; }
    if (txmax < 0 || tymax < 0 || tzmax < 0) return 0;
    if (txmin <= 0 && tymin <= 0 && tzmin <= 0) return 1;
    if (dmax3(txmin, tymin, tzmin) <= dmin3(txmax, tymax, tzmax)) return 1;
    else return 0;
}

/* intersect(Ray&, Face&, Vertex&), MTPC/sceneManagement.cpp:316-338 */
static int tri_hit(const RayT* r, const Face* tr, vec3* ret)
{
    vec3 norm = tr->norm;
    double t = vdot(vsub(tr->v1, r->o), norm) / vdot(norm, r->d);
    vec3 p = vadd(r->o, vmul(r->d, t));
    vec3 ap = vsub(p, tr->v1), bp = vsub(p, tr->v2), cp = vsub(p, tr->v3);
    vec3 ab = vsub(tr->v2, tr->v1), bc = vsub(tr->v3, tr->v2), ca = vsub(tr->v1, tr->v3);
    vec3 cross1 = vcross(ab, ap), cross2 = vcross(bc, bp), cross3 = vcross(ca, cp);
    double dir1 = vdot(cross1, norm), dir2 = vdot(cross2, norm), dir3 = vdot(cross3, norm);
    double j1 = dir1 * dir2, j2 = dir1 * dir3, j3 = dir2 * dir3;
    int judge = 0;
    if (j1 >= 0 && j2 >= 0 && j3 >= 0) judge = 1;
    *ret = p;
    return judge;
}

/* findGarCor, MTPC/pathTracing.cpp:394-432 ("plan 2") */
static vec3 barycentric(const Face* f, vec3 p)
{
    vec3 e1 = vsub(f->v3, f->v2), e2 = vsub(f->v1, f->v3), e3 = vsub(f->v2, f->v1);
    vec3 d1 = vsub(p, f->v1), d2 = vsub(p, f->v2), d3 = vsub(p, f->v3);
    vec3 n = vcross(e1, e2);
    double an = vdot(n, n);
    double b1 = vdot(vcross(e1, d3), n) / an;
    double b2 = vdot(vcross(e2, d1), n) / an;
    double b3 = vdot(vcross(e3, d2), n) / an;
    return v3(b1, b2, b3);
}

typedef struct {
    int hit;                 /* flag */
    int leaf;                /* leaf (sorted) index of i.f */
    double t; vec3 p, pn;
} Hit;

typedef struct { uint64_t box, tri; } Cnt;

static inline void leaf_visit(const orc_scene* s, const RayT* r, int leaf, Hit* v, Cnt* c)
{
    /* pathTracing.cpp:340-362 */
    const Face* f = &s->f[s->order[leaf]];
    vec3 ret;
    c->tri++;
    if (tri_hit(r, f, &ret)) {
        double t = (ret.x - r->o.x) / r->d.x;                   /* :347 */
        vec3 g = barycentric(f, ret);
        vec3 pn = vadd(vadd(vmul(f->vn1, g.x), vmul(f->vn2, g.y)), vmul(f->vn3, g.z));   /* :351 */
        if (!v->hit) {
            if (t > 0) { v->hit = 1; v->leaf = leaf; v->t = t; v->p = ret; v->pn = pn; }
        } else if (t > 0 && t < v->t) { v->leaf = leaf; v->t = t; v->p = ret; v->pn = pn; }
    }
}

/* bvh_intersect with virtual children skipped (D5), MTPC/pathTracing.cpp:334-374 */
static void bvh_walk(const orc_scene* s, const RayT* r, Hit* v, int node, int level, Cnt* c)
{
    const orc_bvh_info* b = &s->bi;
    const Node* n = &s->nodes[find_index(b, node, level)];
    c->box++;
    if (box_hit(r, n->b)) {
        if (n->level == b->Level) { leaf_visit(s, r, n->leaf, v, c); return; }
        bvh_walk(s, r, v, 2 * node + 1, level + 1, c);
        if (have_right(b, node, level)) bvh_walk(s, r, v, 2 * node + 2, level + 1, c);
    }
}

/* bvh_intersect exactly as written, aliasing included (Q7).  index >= Nr (odd t) is skipped. */
static void bvh_walk_alias(const orc_scene* s, const RayT* r, Hit* v, int node, int level, Cnt* c)
{
    const orc_bvh_info* b = &s->bi;
    int index = find_index(b, node, level);
    if (index < 0 || index >= b->Nr) return;
    const Node* n = &s->nodes[index];
    c->box++;
    if (box_hit(r, n->b)) {
        if (n->level == b->Level) { leaf_visit(s, r, n->leaf, v, c); return; }
        bvh_walk_alias(s, r, v, 2 * node + 1, level + 1, c);
        bvh_walk_alias(s, r, v, 2 * node + 2, level + 1, c);
    }
}

static void flat_walk(const orc_scene* s, const RayT* r, Hit* v, Cnt* c)
{
    const orc_bvh_info* b = &s->bi;
    int first = find_index(b, (1 << b->Level) - 1, b->Level);
    for (int k = 0; k < b->t; k++) {
        c->box++;
        if (box_hit(r, s->nodes[first + k].b)) leaf_visit(s, r, k, v, c);
    }
}

/* ray_intersect, MTPC/pathTracing.cpp:382-390 */
static int ray_intersect(const orc_scene* s, const RayT* r, Hit* v, int mode, Cnt* c)
{
    v->hit = 0; v->leaf = -1; v->t = 0; v->p = v3(0, 0, 0); v->pn = v3(0, 0, 0);
    if (mode == ORC_TRACE_ALIAS) bvh_walk_alias(s, r, v, 0, 0, c);
    else if (mode == ORC_TRACE_FLAT) flat_walk(s, r, v, c);
    else bvh_walk(s, r, v, 0, 0, c);
    return v->hit;
}

void orc_trace_closest(const orc_scene* s, const double* rays, int64_t n, int mode,
                       int32_t* face, double* t, double* p, double* pn, orc_stats* st)
{
    uint64_t tb = 0, tt = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : tb, tt)
    for (int64_t i = 0; i < n; i++) {
        RayT r = { v3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), v3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]) };
        Hit h; Cnt c = { 0, 0 };
        int ok = ray_intersect(s, &r, &h, mode, &c);
        tb += c.box; tt += c.tri;
        if (face) face[i] = ok ? s->order[h.leaf] : -1;
        if (t) t[i] = ok ? h.t : 0.0;
        if (p) { p[i * 3] = h.p.x; p[i * 3 + 1] = h.p.y; p[i * 3 + 2] = h.p.z; }
        if (pn) { pn[i * 3] = h.pn.x; pn[i * 3 + 1] = h.pn.y; pn[i * 3 + 2] = h.pn.z; }
    }
    if (st) { st->box_tests += tb; st->tri_tests += tt; }
}

/* ------------------------------------------------------------------ integrator */
enum { RT_DIFFUSE = 0, RT_SPECULAR = 1, RT_TRANSMISSION = 2 };   /* sceneManagement.h:203-205 */

typedef struct {
    const orc_scene* s;
    uint64_t seed; uint32_t pixel, sample;
    int faithful_cost;
    orc_stats* st; Cnt c;
} Ctx;

/* slot table per path vertex: light i -> 4i..4i+3, then RR, FRESNEL, LOBE, PHI, THETA */
#define SLOT_RR(nl)      (4u * (uint32_t)(nl) + 0u)
#define SLOT_FRESNEL(nl) (4u * (uint32_t)(nl) + 1u)
#define SLOT_LOBE(nl)    (4u * (uint32_t)(nl) + 2u)
#define SLOT_PHI(nl)     (4u * (uint32_t)(nl) + 3u)
#define SLOT_THETA(nl)   (4u * (uint32_t)(nl) + 4u)

static inline double U(const Ctx* c, int depth, uint32_t slot)
{
    return orc_uniform(c->seed, c->pixel, c->sample, (uint32_t)depth, slot);
}

/* Refract, MTPC/pathTracing.cpp:13-27 (float cosi / cost2) */
static int refract_dir(vec3 i, vec3 n, double eta, vec3* out)
{
    float cosi = (float)vdot(i, n);
    float cost2 = (float)(1.0f - eta * eta * (1.0f - cosi * cosi));
    if (cost2 >= 0.0f) {
        *out = vsub(vmul(i, eta), vmul(n, (eta * cosi + sqrtf(cost2))));
        return 1;
    }
    return 0;
}

/* BRDFImportanceSampling, MTPC/pathTracing.cpp:30-64 */
static vec3 brdf_sample(const Ctx* c, int depth, vec3 direction, int type, double Ns)
{
    int nl = c->s->nl;
    double phi = U(c, depth, SLOT_PHI(nl)) * 2 * ORC_PI;
    double theta;
    if (type == RT_DIFFUSE) theta = asin(sqrt(U(c, depth, SLOT_THETA(nl))));
    else theta = acos(pow(U(c, depth, SLOT_THETA(nl)), (double)1 / (Ns + 1)));
    vec3 sample = v3(sin(theta) * cos(phi), cos(theta), sin(theta) * sin(phi));
    vec3 front;
    if (fabs(direction.x) > fabs(direction.y)) front = vnormalize(v3(direction.z, 0, -direction.x));
    else front = vnormalize(v3(0, -direction.z, direction.y));
    vec3 right = vcross(direction, front);
    vec3 ret = vadd(vadd(vmul(right, sample.x), vmul(direction, sample.y)), vmul(front, sample.z));
    return vnormalize(ret);
}

typedef struct { vec3 o, d; int type; } NextRay;

/* nextRay, MTPC/pathTracing.cpp:66-134 */
static NextRay next_ray(const Ctx* c, int depth, const Hit* p, const Material* m, vec3 dir, vec3 kd)
{
    int nl = c->s->nl;
    NextRay out;
    vec3 direction;
    if (m->Ni > 1) {
        double n1, n2;
        double cos_in = vdot(vneg(dir), p->pn);
        vec3 normal;
        if (cos_in > 0) { normal = vneg(p->pn); n1 = m->Ni; n2 = 1.0; }
        else { normal = p->pn; n1 = 1.0; n2 = m->Ni; }
        double rf0 = pow((n1 - n2) / (n1 + n2), 2);
        double fresnel = rf0 + (1.0f - rf0) * pow(1.0f - fabs(cos_in), 5);
        if (fresnel < U(c, depth, SLOT_FRESNEL(nl))) {
            if (refract_dir(vneg(dir), normal, n1 / n2, &direction)) {
                out.o = p->p; out.d = direction; out.type = RT_TRANSMISSION;
                if (c->st) c->st->rays_on_surface++;
                return out;
            } else {
                vec3 incoming = vneg(dir);
                vec3 reflect = vsub(incoming, vmul(vmul(normal, vdot(incoming, normal)), 2));
                out.o = p->p; out.d = reflect; out.type = RT_SPECULAR;
                if (c->st) c->st->rays_on_surface++;
                return out;
            }
        }
    }
    double kd_norm = vnorm(kd), ks_norm = vnorm(m->ks);
    int type;
    if (ks_norm != 0 && kd_norm / ks_norm < U(c, depth, SLOT_LOBE(nl))) {
        vec3 incoming = vneg(dir);
        vec3 reflect = vsub(incoming, vmul(vmul(p->pn, vdot(incoming, p->pn)), 2));
        direction = brdf_sample(c, depth, reflect, RT_SPECULAR, m->Ns);
        type = RT_SPECULAR;
    } else {
        direction = brdf_sample(c, depth, p->pn, RT_DIFFUSE, m->Ns);
        type = RT_DIFFUSE;
    }
    out.o = vadd(p->p, vmul(direction, 0.01));
    out.d = direction; out.type = type;
    return out;
}


/* shade, MTPC/pathTracing.cpp:137-266.  p->leaf identifies p.f. */
static vec3 shade(Ctx* c, const Hit* p, vec3 dir, int depth)
{
    const orc_scene* s = c->s;
    const Face* pf = &s->f[s->order[p->leaf]];
    const Material* m = &s->m[pf->material];
    const int nl = s->nl;
    if (c->st) { c->st->shade_calls++; if (depth > c->st->max_depth) c->st->max_depth = depth; }
    if (m->light >= 0) return s->l[m->light].radiance;             /* :141-144 */

    vec3 kd;
    if (m->has_map) {                                               /* :147-160, Q9 */
        vec3 g = barycentric(pf, p->p);
        double row = pf->vt1[0] * g.x + pf->vt2[0] * g.y + pf->vt3[0] * g.z;
        double col = pf->vt1[1] * g.x + pf->vt2[1] * g.y + pf->vt3[1] * g.z;
        double irow = row - floor(row), icol = col - floor(col);
        int r = (int)(irow * m->map_h), cc = (int)(icol * m->map_w);
        if (r < 0) r = 0; if (r > m->map_h - 1) r = m->map_h - 1;          /* D7 */
        if (cc < 0) cc = 0; if (cc > m->map_w - 1) cc = m->map_w - 1;
        const uint8_t* px = m->bgr + ((size_t)r * m->map_w + cc) * 3;
        kd = v3((double)px[2] / 255, (double)px[1] / 255, (double)px[0] / 255);
    } else kd = m->kd;

    /* direct illumination, :166-232 */
    vec3 L_dir = v3(0, 0, 0);
    int sample_mat = -1;                 /* Face sample_face; (material "" until a triangle is chosen) */
    for (int i = 0; i < nl; i++) {
        const Light* L = &s->l[i];
        const Material* lm = &s->m[L->material];
        vec3 xl = v3(0, 0, 0), vn = v3(0, 0, 0);
        double total_aera; const double* cdf = L->cdf; double* tmp = NULL;
        if (c->faithful_cost) {          /* rebuild the CDF per call like :177-184 (same values) */
            tmp = (double*)malloc(sizeof(double) * (lm->nf ? lm->nf : 1));
            double tot = 0;
            for (int j = 0; j < lm->nf; j++) { tot += face_area(&s->f[lm->faces[j]]); tmp[j] = tot; }
            total_aera = tot; cdf = tmp;
        } else total_aera = L->total_area;
        /* static u1(0, total_aera of the FIRST light ever processed) -- Q1 */
        double rnd = U(c, depth, 4u * i + 0u) * s->area0;
        for (int j = 0; j < lm->nf; j++) {
            if (rnd < cdf[j]) {
                const Face* sf = &s->f[lm->faces[j]];
                sample_mat = sf->material;
                double rnd1 = U(c, depth, 4u * i + 1u), rnd2 = U(c, depth, 4u * i + 2u), rnd3 = U(c, depth, 4u * i + 3u);
                double p1 = rnd1 / (rnd1 + rnd2 + rnd3), p2 = rnd2 / (rnd1 + rnd2 + rnd3), p3 = rnd3 / (rnd1 + rnd2 + rnd3);
                xl = vadd(vadd(vmul(sf->v1, p1), vmul(sf->v2, p2)), vmul(sf->v3, p3));
                vn = vadd(vadd(vmul(sf->vn1, p1), vmul(sf->vn2, p2)), vmul(sf->vn3, p3));
                break;
            }
        }
        free(tmp);
        vec3 direction = vnormalize(vsub(xl, p->p));
        double visibility = 1;
        RayT rl = { vadd(p->p, vmul(direction, 0.01)), direction };
        Hit inter;
        ray_intersect(s, &rl, &inter, s->walk_mode, &c->c);
        if (c->st) c->st->rays_shadow++;
        int inter_mat = inter.hit ? s->f[s->order[inter.leaf]].material : -1;
        if (inter_mat != sample_mat) visibility = 0;                           /* :213 */
        if (vdot(direction, p->pn) > 0) {
            double pdf_light = (double)1 / total_aera;
            double cos_theta = fabs(vdot(direction, vn) / vnorm(direction) / vnorm(vn));
            double cos_theta_hat = fabs(vdot(direction, p->pn) / vnorm(direction) / vnorm(p->pn));
            double dist = std_max(1.0, vnorm(vsub(xl, p->p)));
            vec3 intensity = vmul(vdiv(vdiv(vmul(vmul(L->radiance, cos_theta), cos_theta_hat), pow(dist, 2)), pdf_light), visibility);
            double kd_dots = vdot(direction, p->pn);
            if (kd_dots > 0) {
                L_dir.x += kd.x * intensity.x * kd_dots / ORC_PI;
                L_dir.y += kd.y * intensity.y * kd_dots / ORC_PI;
                L_dir.z += kd.z * intensity.z * kd_dots / ORC_PI;
            }
        }
    }

    /* indirect illumination, :234-263 */
    vec3 L_indir = v3(0, 0, 0);
    double P_RR = ORC_P_RR;
    if (depth + 1 < ORC_MAX_DEPTH && U(c, depth, SLOT_RR(nl)) < P_RR) {         /* russian_Roulette :3-11, D6 */
        NextRay r = next_ray(c, depth, p, m, dir, kd);
        RayT rr = { r.o, r.d };
        Hit ret;
        if (c->st) c->st->rays_bounce++;
        if (ray_intersect(s, &rr, &ret, s->walk_mode, &c->c)) {
            vec3 intensity = vdiv(shade(c, &ret, vneg(r.d), depth + 1), P_RR);
            if (r.type == RT_DIFFUSE) {
                const Material* hm = &s->m[s->f[s->order[ret.leaf]].material];
                if (hm->light < 0) {
                    L_indir.x += kd.x * intensity.x;
                    L_indir.y += kd.y * intensity.y;
                    L_indir.z += kd.z * intensity.z;
                }
            } else if (r.type == RT_SPECULAR) {
                L_indir.x += m->ks.x * intensity.x;
                L_indir.y += m->ks.y * intensity.y;
                L_indir.z += m->ks.z * intensity.z;
            } else {
                L_indir = vadd(L_indir, intensity);
            }
        }
    }
    return vadd(L_dir, L_indir);
}

/* camera frame of generateImg, MTPC/pathTracing.cpp:276-294 */
typedef struct { vec3 eye, start_point, screen_pdx, screen_pdy; } CamFrame;

static CamFrame cam_frame(const orc_scene* s)
{
    CamFrame cf;
    vec3 up = vnormalize(s->up);                                     /* :276 */
    vec3 dir = vsub(s->look_at, s->eye);
    double l = vnorm(dir);
    double dy = tan(s->fovy / 2 / 180 * ORC_PI) * l;
    double dx = dy / s->height * s->width;
    vec3 screen_center = s->look_at;
    double pdx = 2 * dx / s->width, pdy = 2 * dy / s->height;
    vec3 screen_x_dir = vnormalize(vcross(dir, up));
    vec3 screen_y_dir = up;
    cf.screen_pdy = vmul(screen_y_dir, pdy);
    cf.screen_pdx = vmul(screen_x_dir, pdx);
    cf.start_point = vadd(vsub(screen_center, vmul(screen_x_dir, dx)), vmul(up, dy));
    cf.eye = s->eye;
    return cf;
}

/* pixel position: row start then a running sum along the row (:297,:326, Q11) */
static vec3 pixel_pos(const CamFrame* cf, int row, int col)
{
    vec3 pos = vsub(cf->start_point, vmul(cf->screen_pdy, row));
    for (int j = 0; j < col; j++) pos = vadd(pos, cf->screen_pdx);
    return pos;
}

void orc_primary_ray(const orc_scene* s, int row, int col, double ray6[6])
{
    CamFrame cf = cam_frame(s);
    vec3 pos = pixel_pos(&cf, row, col);
    vec3 d = vnormalize(vsub(pos, cf.eye));
    ray6[0] = cf.eye.x; ray6[1] = cf.eye.y; ray6[2] = cf.eye.z; ray6[3] = d.x; ray6[4] = d.y; ray6[5] = d.z;
}

/* the primary rays of rows [row0,row1), every column, in generateImg's own order (one running sum per row) */
void orc_primary_rays(const orc_scene* s, int row0, int row1, double* rays6)
{
    CamFrame cf = cam_frame(s);
    const int W = s->width;
    for (int i = row0; i < row1; i++) {
        vec3 pos = vsub(cf.start_point, vmul(cf.screen_pdy, i));        /* :297 */
        for (int j = 0; j < W; j++) {
            vec3 d = vnormalize(vsub(pos, cf.eye));                     /* :306-308 */
            double* r = rays6 + ((size_t)(i - row0) * W + j) * 6;
            r[0] = cf.eye.x; r[1] = cf.eye.y; r[2] = cf.eye.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
            pos = vadd(pos, cf.screen_pdx);                             /* :326 */
        }
    }
}

static vec3 sample_radiance(Ctx* c, const RayT* ray, const Hit* primary, int have_primary)
{
    Hit h;
    if (have_primary) h = *primary;
    else { if (c->st) c->st->rays_primary++; ray_intersect(c->s, ray, &h, c->s->walk_mode, &c->c); }
    if (c->st) c->st->samples++;
    if (!h.hit) return v3(0, 0, 0);
    return shade(c, &h, vneg(ray->d), 0);
}

void orc_sample_radiance(const orc_scene* s, uint64_t seed, int row, int col, int k, double rgb[3], orc_stats* st)
{
    CamFrame cf = cam_frame(s);
    vec3 pos = pixel_pos(&cf, row, col);
    RayT ray = { cf.eye, vnormalize(vsub(pos, cf.eye)) };
    Ctx c = { s, seed, (uint32_t)(row * s->width + col), (uint32_t)k, 0, st, { 0, 0 } };
    vec3 r = sample_radiance(&c, &ray, NULL, 0);
    if (st) { st->box_tests += c.c.box; st->tri_tests += c.c.tri; }
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}

static void stats_add(orc_stats* a, const orc_stats* b)
{
    a->rays_primary += b->rays_primary; a->rays_shadow += b->rays_shadow; a->rays_bounce += b->rays_bounce;
    a->box_tests += b->box_tests; a->tri_tests += b->tri_tests; a->shade_calls += b->shade_calls; a->samples += b->samples;
    if (b->max_depth > a->max_depth) a->max_depth = b->max_depth;
    a->rays_on_surface += b->rays_on_surface;
}

/* generateImg, MTPC/pathTracing.cpp:274-331 (D1, D3) */
void orc_render(const orc_scene* s, int spp, uint64_t seed, int row0, int row1, int col0, int col1,
                int faithful_cost, int nthreads, double* img, orc_stats* st)
{
    CamFrame cf = cam_frame(s);
    const int W = s->width;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    orc_stats total; memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        orc_stats local; memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1) collapse(1)
        for (int i = row0; i < row1; i++) {
            vec3 pos = vsub(cf.start_point, vmul(cf.screen_pdy, i));
            for (int j = 0; j < col1; j++) {
                if (j >= col0) {
                    float cr = 0, cg = 0, cb = 0;                      /* glm::vec3 current_radiance :301 */
                    RayT ray = { cf.eye, vnormalize(vsub(pos, cf.eye)) };
                    Ctx c = { s, seed, (uint32_t)(i * W + j), 0, faithful_cost, &local, { 0, 0 } };
                    Hit primary; int have = 0;
                    if (!faithful_cost) {                              /* identical for every k: trace once */
                        local.rays_primary++;
                        ray_intersect(s, &ray, &primary, s->walk_mode, &c.c);
                        have = 1;
                    }
                    for (int k = 0; k < spp; k++) {
                        c.sample = (uint32_t)k;
                        if (have && !primary.hit) { local.samples++; continue; }
                        vec3 radiance = sample_radiance(&c, &ray, &primary, have);
                        if (!have) {
                            /* faithful: only hits accumulate (:311-319); a miss adds nothing */
                        }
                        cr += radiance.x / spp;                         /* float += double (:316-318) */
                        cg += radiance.y / spp;
                        cb += radiance.z / spp;
                    }
                    local.box_tests += c.c.box; local.tri_tests += c.c.tri;
                    img[((size_t)i * W + j) * 3 + 0] = cr;
                    img[((size_t)i * W + j) * 3 + 1] = cg;
                    img[((size_t)i * W + j) * 3 + 2] = cb;
                }
                pos = vadd(pos, cf.screen_pdx);                         /* :326 */
            }
        }
#pragma omp critical
        stats_add(&total, &local);
    }
    if (st) stats_add(st, &total);
}

/* The reference's own parallel structure, for timing (MTPC/pathTracing.cpp:300-320): the samples of ONE pixel at a time on
 * min(spp, 8) OpenMP threads -- a team forked and joined per pixel, as `omp_set_num_threads(min(N, 8)); #pragma omp parallel for`
 * over k does there -- every sample re-tracing the primary ray and rebuilding the light CDF per shade call (faithful cost).  The
 * reference adds the samples under a mutex in whatever order the threads arrive; here they are summed in sample order after the join
 * (D3), so the picture equals orc_render's.  Pixels: rows 0, row_stride, ... x columns 0, col_stride, ... */
void orc_render_reference_style(const orc_scene* s, int spp, uint64_t seed, int row_stride, int col_stride, double* img, orc_stats* st)
{
    CamFrame cf = cam_frame(s);
    const int W = s->width, H = s->height;
    const int team = spp < 8 ? spp : 8;
    vec3* rad = (vec3*)malloc(sizeof(vec3) * (size_t)spp);
    orc_stats total; memset(&total, 0, sizeof total);
    for (int i = 0; i < H; i++) {
        vec3 pos = vsub(cf.start_point, vmul(cf.screen_pdy, i));
        for (int j = 0; j < W; j++) {
            if (i % row_stride == 0 && j % col_stride == 0) {
                const RayT ray = { cf.eye, vnormalize(vsub(pos, cf.eye)) };
#ifdef _OPENMP
                omp_set_num_threads(team);
#endif
#pragma omp parallel
                {
                    orc_stats local; memset(&local, 0, sizeof local);
                    uint64_t box = 0, tri = 0;
#pragma omp for schedule(static)
                    for (int k = 0; k < spp; k++) {
                        Ctx c = { s, seed, (uint32_t)(i * W + j), (uint32_t)k, 1, &local, { 0, 0 } };
                        RayT r = ray;
                        Hit primary;
                        rad[k] = sample_radiance(&c, &r, &primary, 0);
                        box += c.c.box; tri += c.c.tri;
                    }
                    local.box_tests += box; local.tri_tests += tri;
#pragma omp critical
                    stats_add(&total, &local);
                }
                float cr = 0, cg = 0, cb = 0;
                for (int k = 0; k < spp; k++) { cr += rad[k].x / spp; cg += rad[k].y / spp; cb += rad[k].z / spp; }
                img[((size_t)i * W + j) * 3 + 0] = cr; img[((size_t)i * W + j) * 3 + 1] = cg; img[((size_t)i * W + j) * 3 + 2] = cb;
            }
            pos = vadd(pos, cf.screen_pdx);
        }
    }
    free(rad);
    if (st) stats_add(st, &total);
}

/* imshow, MTPC/MTPC.cpp:22-30: (unsigned char)clamp(v*255, 0, 255) */
void orc_quantize(const double* img, int64_t n, uint8_t* rgb8)
{
    for (int64_t i = 0; i < n; i++) {
        double v = img[i] * 255;
        v = (v < 0.0) ? 0.0 : v;           /* glm::clamp = min(max(x,lo),hi) */
        v = (255.0 < v) ? 255.0 : v;
        rgb8[i] = (uint8_t)v;
    }
}

/* Uncompressed PNG exactly as svpng emits it (MTPC/svpng.inc:77-107): 8-bit RGB, zlib header 78 01,
 * one stored deflate block per row, filter byte 0, adler32, no ancillary chunks. */
static uint32_t crc_table[256]; static int crc_ready = 0;
static void crc_init(void)
{
    for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; }
    crc_ready = 1;
}
typedef struct { uint8_t* out; int64_t pos, cap; uint32_t crc; int ovf; } PngW;
static void put(PngW* w, uint8_t b) { if (w->pos < w->cap) w->out[w->pos] = b; else w->ovf = 1; w->pos++; }
static void putc_crc(PngW* w, uint8_t b) { put(w, b); w->crc = crc_table[(w->crc ^ b) & 255] ^ (w->crc >> 8); }
static void put32(PngW* w, uint32_t u) { put(w, u >> 24); put(w, (u >> 16) & 255); put(w, (u >> 8) & 255); put(w, u & 255); }
static void put32_crc(PngW* w, uint32_t u) { putc_crc(w, u >> 24); putc_crc(w, (u >> 16) & 255); putc_crc(w, (u >> 8) & 255); putc_crc(w, u & 255); }
static void begin(PngW* w, const char* tag, uint32_t len) { put32(w, len); w->crc = ~0u; for (int i = 0; i < 4; i++) putc_crc(w, (uint8_t)tag[i]); }
static void end(PngW* w) { put32(w, ~w->crc); }

int64_t orc_png_encode(const uint8_t* img, int w, int h, uint8_t* out, int64_t cap)
{
    if (!crc_ready) crc_init();
    PngW pw = { out, 0, cap, 0, 0 };
    static const uint8_t magic[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    for (int i = 0; i < 8; i++) put(&pw, magic[i]);
    begin(&pw, "IHDR", 13);
    put32_crc(&pw, (uint32_t)w); put32_crc(&pw, (uint32_t)h);
    putc_crc(&pw, 8); putc_crc(&pw, 2); putc_crc(&pw, 0); putc_crc(&pw, 0); putc_crc(&pw, 0);
    end(&pw);
    uint32_t p = (uint32_t)w * 3 + 1;
    begin(&pw, "IDAT", 2 + (uint32_t)h * (5 + p) + 4);
    putc_crc(&pw, 0x78); putc_crc(&pw, 0x01);
    uint32_t a = 1, b = 0;
    for (int y = 0; y < h; y++) {
        putc_crc(&pw, y == h - 1);
        putc_crc(&pw, p & 255); putc_crc(&pw, (p >> 8) & 255);
        putc_crc(&pw, (~p) & 255); putc_crc(&pw, ((~p) >> 8) & 255);
        putc_crc(&pw, 0); a = (a + 0) % 65521; b = (b + a) % 65521;
        for (uint32_t x = 0; x < p - 1; x++) {
            uint8_t u = *img++;
            putc_crc(&pw, u); a = (a + u) % 65521; b = (b + a) % 65521;
        }
    }
    put32_crc(&pw, (b << 16) | a);
    end(&pw);
    begin(&pw, "IEND", 0); end(&pw);
    return pw.ovf ? -1 : pw.pos;
}

/* Same as orc_render for rows 0, stride, 2*stride, ... (all columns): the bounded CPU-baseline sample.
 * Work items are (sampled row, 64-column block) pairs handed out dynamically to the OpenMP team. */
void orc_render_strided(const orc_scene* s, int spp, uint64_t seed, int row_stride, int faithful_cost, int nthreads,
                        int unused, double* img, orc_stats* st)
{
    (void)unused;
    const int nrows = (s->height + row_stride - 1) / row_stride;
    const int cb = 64, ncb = (s->width + cb - 1) / cb;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    orc_stats total; memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        orc_stats local; memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
        for (int w = 0; w < nrows * ncb; w++) {
            const int r = (w / ncb) * row_stride, c0 = (w % ncb) * cb;
            const int c1 = c0 + cb < s->width ? c0 + cb : s->width;
            orc_render(s, spp, seed, r, r + 1, c0, c1, faithful_cost, -1, img, &local);
        }
#pragma omp critical
        stats_add(&total, &local);
    }
    if (st) stats_add(st, &total);
}
