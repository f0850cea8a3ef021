// Host-side logic of liborbx that needs no GPU: constructor tables, pyramid/cell geometry, resize
// coefficient tables, the frame grid, scalar matcher helpers and the synthetic input generators.
// Compiled with -ffp-contract=off: every float expression rounds per operation, as the reference's
// C++ does on a baseline x86-64 build.
#include "orbx_internal.h"
#include <cmath>
#include <cfloat>
#include <cstring>
#include <climits>
#include <algorithm>

namespace orbx {

static thread_local std::string g_last_error;
void set_error(const std::string& s) { g_last_error = s; }

static inline int cv_round(double v) { return (int)std::lrint(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
static inline short sat_short(int v) { return (short)std::min(std::max(v, (int)SHRT_MIN), (int)SHRT_MAX); }

// ORBextractor::ORBextractor, src/ORBextractor.cc:409-469
int build_scale_tables(const orbx_params& p, ScaleTables& t)
{
    if (p.nlevels < 1 || p.nlevels > kMaxLevels || p.nfeatures < 0 || !(p.scale_factor > 1.0f)) {
        set_error("orbx_params: need 1 <= nlevels <= 32, nfeatures >= 0, scale_factor > 1");
        return ORBX_ERR_BAD_ARG;
    }
    if (p.trig_mode < 0 || p.trig_mode > 2 || p.blur_rounding < 0 || p.blur_rounding > 1) {
        set_error("orbx_params: blur_rounding is 0 or 1, trig_mode 0, 1 or 2");
        return ORBX_ERR_BAD_ARG;
    }
    const double scaleFactor = p.scale_factor; // the member is a double (include/ORBextractor.h:223)
    t.nlevels = p.nlevels;
    t.scale[0] = 1.0f;
    t.sigma2[0] = 1.0f;
    for (int i = 1; i < p.nlevels; ++i) {
        t.scale[i] = (float)(t.scale[i - 1] * scaleFactor);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    for (int i = 0; i < p.nlevels; ++i) {
        t.inv_scale[i] = 1.0f / t.scale[i];
        t.inv_sigma2[i] = 1.0f / t.sigma2[i];
    }
    const float factor = (float)(1.0f / scaleFactor);
    float desired = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
    int acc = 0;
    for (int l = 0; l + 1 < p.nlevels; ++l) {
        t.quota[l] = cv_round(desired);
        acc += t.quota[l];
        desired *= factor;
    }
    t.quota[p.nlevels - 1] = std::max(p.nfeatures - acc, 0);

    // circular patch row ends, :453-468
    int um[kHalfPatch + 2] = {0};
    const int vmax = cv_floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = cv_ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) um[v] = cv_round(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (um[v0] == um[v0 + 1]) ++v0;
        um[v] = v0;
        ++v0;
    }
    for (int v = 0; v < 16; ++v) t.umax[v] = um[v];
    return ORBX_OK;
}

// Coefficients of cv::resize(INTER_LINEAR) for 8-bit images (OpenCV imgproc, generic path):
// Q11 weights, source offsets clamped at the right/bottom edge.
static void build_resize_tables(int sw, int sh, int dw, int dh, std::vector<ResizeX>& rx, std::vector<ResizeY>& ry, int& area2x)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    const int iscale_x = cv_round(scale_x), iscale_y = cv_round(scale_y);
    const bool area_fast = std::abs(scale_x - iscale_x) < DBL_EPSILON && std::abs(scale_y - iscale_y) < DBL_EPSILON;
    area2x = (area_fast && iscale_x == 2 && iscale_y == 2) ? 1 : 0;
    for (int dx = 0; dx < dw; ++dx) {
        ResizeX e;
        if (area2x) { e.sx = 2 * dx; e.a0 = e.a1 = 0; rx.push_back(e); continue; }
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        e.sx = sx;
        e.a0 = sat_short(cv_round((1.f - fx) * 2048));
        e.a1 = sat_short(cv_round(fx * 2048));
        rx.push_back(e);
    }
    for (int dy = 0; dy < dh; ++dy) {
        ResizeY e;
        if (area2x) { e.sy0 = (int16_t)(2 * dy); e.sy1 = (int16_t)(2 * dy + 1); e.b0 = e.b1 = 0; ry.push_back(e); continue; }
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        e.b0 = sat_short(cv_round((1.f - fy) * 2048));
        e.b1 = sat_short(cv_round(fy * 2048));
        e.sy0 = (int16_t)std::min(std::max(sy, 0), sh - 1);
        e.sy1 = (int16_t)std::min(std::max(sy + 1, 0), sh - 1);
        ry.push_back(e);
    }
}

int build_geometry(const orbx_params& p, int width, int height, Geometry& g)
{
    if (width <= 0 || height <= 0) { set_error("image size must be positive"); return ORBX_ERR_BAD_ARG; }
    int rc = build_scale_tables(p, g.tables);
    if (rc) return rc;
    g.width = width; g.height = height; g.nlevels = p.nlevels;
    g.lv.assign(p.nlevels, LevelGeom());
    g.cells.clear(); g.rx.clear(); g.ry.clear(); g.rrb.clear(); g.rga.clear(); g.rgb.clear();
    size_t off = 0, boff = 0;
    int cand_off = 0, kp_off = 0;
    g.out_cap = 0; g.max_cell_dim = 0;
    const float W = 30; // src/ORBextractor.cc:768
    for (int l = 0; l < p.nlevels; ++l) {
        LevelGeom& L = g.lv[l];
        const float s = g.tables.inv_scale[l];
        L.w = cv_round((float)width * s);      // :1111
        L.h = cv_round((float)height * s);
        L.scale = g.tables.scale[l];
        L.kp_size = (float)(int)(kPatch * g.tables.scale[l]);
        L.quota = g.tables.quota[l];
        L.rows = L.h + 2 * kEdge;
        L.cpr = (L.w + kFrameFill + 15) / 16;   // chunks also cover the frame columns the producers fill
        L.cpr_magic = (unsigned)((0x100000000ull + L.cpr - 1) / L.cpr);
        for (uint32_t i = 0, n = (uint32_t)L.cpr * (uint32_t)L.h; i < n; ++i)     // the reciprocal must be exact over the index range
            if ((uint32_t)(((uint64_t)i * L.cpr_magic) >> 32) != i / (uint32_t)L.cpr) { set_error("internal: chunk reciprocal inexact"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
        L.gpr = 1 + (L.w + kFrameFill + 3) / 4;
        L.gpr_magic = (unsigned)((0x100000000ull + L.gpr - 1) / L.gpr);
        for (uint32_t i = 0, n = (uint32_t)L.gpr * (uint32_t)((L.h + 3) / 4); i < n; ++i)
            if ((uint32_t)(((uint64_t)i * L.gpr_magic) >> 32) != i / (uint32_t)L.gpr) { set_error("internal: group reciprocal inexact"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
        L.stride = (kRoiX + L.w + kEdge + 63) & ~63;
        if (L.stride % 512 == 0) L.stride += 64;   // rows a multiple of 512 bytes apart fall on the same memory channels
        L.offset = off;
        L.roi = off + (size_t)kEdge * L.stride + kRoiX;
        off += ((size_t)L.stride * L.rows + 255) & ~(size_t)255;
        L.tpr = (L.w + 15) / 16 + 2;
        L.upr = (L.w + 15) / 16;
        L.upr_magic = (unsigned)((0x100000000ull + L.upr - 1) / L.upr);
        for (uint32_t i = 0, n = (uint32_t)L.upr * (uint32_t)((L.h + 15) / 16) + 64; i < n; ++i)
            if ((uint32_t)(((uint64_t)i * L.upr_magic) >> 32) != i / (uint32_t)L.upr) { set_error("internal: unit reciprocal inexact"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
        L.blur_offset = boff;
        boff += (size_t)((L.h + 7) / 8) * L.tpr * 128;

        L.maxBorderX = L.w - kEdge + 3;        // :774-775
        L.maxBorderY = L.h - kEdge + 3;
        const float fw = (float)(L.maxBorderX - kMinBorder);
        const float fh = (float)(L.maxBorderY - kMinBorder);
        if (!(fw > 0) || !(fh > 0)) { set_error("pyramid level smaller than the 32-pixel border"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
        L.nCols = (int)(fw / W);               // :783-786
        L.nRows = (int)(fh / W);
        // A level whose detection area is under 30 pixels wide or high has nCols or nRows == 0: the reference's float
        // divisions (:785-786) give inf without trapping, its cell loops run zero times and DistributeOctTree returns
        // nothing for the empty candidate list, so the level simply yields no keypoints while its quota stays unused.
        const bool no_cells = L.nCols <= 0 || L.nRows <= 0;
        if (no_cells) { L.nCols = L.nRows = 0; L.wCell = L.hCell = 0; }
        else {
            L.wCell = (int)std::ceil(fw / L.nCols);
            L.hCell = (int)std::ceil(fh / L.nRows);
        }
        if (L.maxBorderX - kMinBorder - 4 >= 4096 || L.maxBorderY - kMinBorder - 4 >= 4096) {
            set_error("level larger than 4128 pixels: packed 12-bit candidate coordinates overflow");
            return ORBX_ERR_UNSUPPORTED_GEOMETRY;
        }
        // DistributeOctTree roots, :542-544 (never consulted for a level without cells: no key is ever assigned to a root)
        L.nIni = (int)std::round((float)(L.maxBorderX - kMinBorder) / (L.maxBorderY - kMinBorder));
        if (no_cells) L.nIni = std::min(std::max(L.nIni, 1), kMaxRoots);
        if (L.nIni <= 0) { set_error("width/height < 0.5: the reference indexes an empty root list (src/ORBextractor.cc:542-568)"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
        if (L.nIni > kMaxRoots) { set_error("width/height > 64 not supported"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
        L.hX = (float)(L.maxBorderX - kMinBorder) / L.nIni;

        // cells, :788-806
        L.cell_first = (int)g.cells.size();
        L.cand_offset = cand_off;
        int slot = 0;
        for (int i = 0; i < L.nRows; ++i) {
            const float iniY = (float)(kMinBorder + i * L.hCell);
            float maxY = iniY + L.hCell + 6;
            if (iniY >= L.maxBorderY - 3) continue;
            if (maxY > L.maxBorderY) maxY = (float)L.maxBorderY;
            for (int j = 0; j < L.nCols; ++j) {
                const float iniX = (float)(kMinBorder + j * L.wCell);
                float maxX = iniX + L.wCell + 6;
                if (iniX >= L.maxBorderX - 6) continue;
                if (maxX > L.maxBorderX) maxX = (float)L.maxBorderX;
                const int cw = (int)maxX - (int)iniX - 6, ch = (int)maxY - (int)iniY - 6;
                if (cw < 1 || ch < 1) continue; // cv::FAST finds nothing in a sub-image narrower than 7
                CellRec c;
                c.x0 = (int16_t)iniX; c.y0 = (int16_t)iniY; c.cw = (int16_t)cw; c.ch = (int16_t)ch;
                c.offx = (int16_t)(j * L.wCell); c.offy = (int16_t)(i * L.hCell);
                c.level = (int16_t)l; c.pad = 0;
                c.slot = slot;
                c.cap = ((cw + 1) / 2) * ((ch + 1) / 2);
                slot += c.cap;
                g.max_cell_dim = std::max(g.max_cell_dim, std::max(cw, ch));
                g.cells.push_back(c);
            }
        }
        L.cell_count = (int)g.cells.size() - L.cell_first;
        L.cand_cap = slot;
        cand_off += slot;

        L.kp_cap = std::max(L.quota + 3, 4 * L.nIni) + 1;
        L.kp_offset = kp_off;
        kp_off += L.kp_cap;
        g.out_cap += L.kp_cap;

        L.area2x = 0; L.fast_resize = 0; L.rx_offset = (int)g.rx.size(); L.ry_offset = (int)g.ry.size();
        L.rg_offset = (int)g.rga.size(); L.rb_offset = (int)g.rrb.size();
        if (l > 0) {
            build_resize_tables(g.lv[l - 1].w, g.lv[l - 1].h, L.w, L.h, g.rx, g.ry, L.area2x);
            // fast kernel: every aligned group of four destination columns must read <= 8 consecutive source bytes
            // Group records: group 0 is the frame dword left of the ROI, then the ROI and kFrameFill frame columns
            // right of it.  Frame columns repeat their REFLECT_101 mirror column (-j -> j, w-1+j -> w-1-j), so the
            // resize kernel writes the part of the frame that later stages read; columns beyond repeat the last one.
            L.fast_resize = L.area2x ? 0 : 1;
            const size_t mark = g.rga.size();
            for (int gx = 0; gx < L.gpr && L.fast_resize; ++gx) {
                ResizeGroupA ra{}; ResizeGroupB rb{};
                int col[4], lo = 1 << 30, hi = -1;
                for (int k = 0; k < 4; ++k) {
                    int x = std::min(4 * (gx - 1) + k, L.w + kFrameFill - 1);
                    if (x < 0) x = -x;
                    if (x >= L.w) x = 2 * L.w - 2 - x;
                    col[k] = std::min(std::max(x, 0), L.w - 1);
                    lo = std::min(lo, (int)g.rx[L.rx_offset + col[k]].sx);
                    hi = std::max(hi, (int)g.rx[L.rx_offset + col[k]].sx);
                }
                if (hi + 1 - lo > 7) { L.fast_resize = 0; break; }   // both neighbours of every column inside 8 bytes
                ra.sx_first = lo;
                for (int k = 0; k < 4; ++k) {
                    const ResizeX& e = g.rx[L.rx_offset + col[k]];
                    ra.offs |= (uint32_t)(e.sx - lo) << (8 * k);
                    const uint32_t wt = (uint32_t)(uint16_t)e.a0 | ((uint32_t)(uint16_t)e.a1 << 16);
                    if (k < 2) ra.w01[k] = wt; else rb.w23[k - 2] = wt;
                }
                g.rga.push_back(ra); g.rgb.push_back(rb);
            }
            if (!L.fast_resize) { g.rga.resize(mark); g.rgb.resize(mark); }
            // row blocks: four destination rows that step through the source one or two rows at a time
            const int sh = g.lv[l - 1].h;
            const size_t rmark = g.rrb.size();
            for (int y0 = 0; y0 < L.h && L.fast_resize; y0 += 4) {
                const int sbase = g.ry[L.ry_offset + y0].sy0;
                ResizeRowBlock rec{};
                bool fits = sbase < (1 << 21);
                for (int r = 0; r < 4 && fits; ++r) {
                    if (y0 + r >= L.h) break;                       // rows below the image are computed but not stored
                    const ResizeY& e = g.ry[L.ry_offset + y0 + r];
                    const int d = e.sy0 - sbase;
                    fits = (d == r || d == r + 1) && e.sy1 == std::min(e.sy0 + 1, sh - 1) && e.b0 >= 0 && e.b0 < 4096 && e.b1 >= 0 && e.b1 < 4096;
                    rec.w[r] = (uint32_t)e.b0 | ((uint32_t)e.b1 << 12) | ((d == r + 1 ? 1u : 0u) << 24);
                }
                if (!fits) { g.rrb.resize(rmark); break; }
                for (int k = 0; k < 3; ++k) rec.w[k] |= (((uint32_t)sbase >> (7 * k)) & 0x7Fu) << 25;
                g.rrb.push_back(rec);
                if (y0 + 4 >= L.h) L.fast_resize = 2;
            }
        }
    }
    g.pyr_bytes = off;
    g.blur_bytes = boff;
    g.cand_total = cand_off;
    g.kp_total = kp_off;
    return ORBX_OK;
}

// ---- frame grid, src/Frame.cc:230-245, 327-392 ----
constexpr int kGridCols = 64, kGridRows = 48;

void FrameGrid::build(const orbx_keypoint* k, int n_, float minX_, float maxX_, float minY_, float maxY_)
{
    kps = k; n = n_; minX = minX_; maxX = maxX_; minY = minY_; maxY = maxY_;
    invW = (float)kGridCols / (maxX - minX);
    invH = (float)kGridRows / (maxY - minY);
    std::vector<int> cell(n, -1);
    cell_start.assign(kGridCols * kGridRows + 1, 0);
    for (int i = 0; i < n; ++i) {
        const int px = (int)std::round((k[i].x - minX) * invW);   // PosInGrid uses round()
        const int py = (int)std::round((k[i].y - minY) * invH);
        if (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) continue;
        cell[i] = px * kGridRows + py;
        ++cell_start[cell[i] + 1];
    }
    for (size_t c = 1; c < cell_start.size(); ++c) cell_start[c] += cell_start[c - 1];
    cell_items.assign(std::max(n, 1), 0);
    std::vector<int> cursor(cell_start.begin(), cell_start.end() - 1);
    for (int i = 0; i < n; ++i) if (cell[i] >= 0) cell_items[cursor[cell[i]]++] = i;
}

void FrameGrid::query(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const
{
    out.clear();
    const int cx0 = std::max(0, (int)std::floor((x - minX - r) * invW));
    if (cx0 >= kGridCols) return;
    const int cx1 = std::min(kGridCols - 1, (int)std::ceil((x - minX + r) * invW));
    if (cx1 < 0) return;
    const int cy0 = std::max(0, (int)std::floor((y - minY - r) * invH));
    if (cy0 >= kGridRows) return;
    const int cy1 = std::min(kGridRows - 1, (int)std::ceil((y - minY + r) * invH));
    if (cy1 < 0) return;
    const bool check = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = cx0; ix <= cx1; ++ix)
        for (int iy = cy0; iy <= cy1; ++iy) {
            const int c = ix * kGridRows + iy;
            for (int j = cell_start[c]; j < cell_start[c + 1]; ++j) {
                const orbx_keypoint& kp = kps[cell_items[j]];
                if (check) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                if (std::fabs(kp.x - x) < r && std::fabs(kp.y - y) < r) out.push_back(cell_items[j]);
            }
        }
}

// ---- PCG32 (O'Neill), used only by the synthetic generators ----
struct Pcg32 {
    uint64_t state, inc;
    explicit Pcg32(uint64_t seed, uint64_t seq = 54u) { state = 0; inc = (seq << 1u) | 1u; next(); state += seed; next(); }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    uint32_t below(uint32_t n) { return next() % n; }
};

static void paint(uint8_t* img, int w, int h, ptrdiff_t stride, int x0, int y0, int rw, int rh, uint8_t v)
{
    const int xa = std::max(x0, 0), xb = std::min(x0 + rw, w), ya = std::max(y0, 0), yb = std::min(y0 + rh, h);
    for (int y = ya; y < yb; ++y)
        for (int x = xa; x < xb; ++x) img[y * stride + x] = v;
}

static void synth(uint32_t frame_index, int w, int h, uint8_t* left, uint8_t* right, ptrdiff_t stride)
{
    Pcg32 rng(0x5EED0000ull + frame_index);
    for (int y = 0; y < h; ++y) {
        std::memset(left + y * stride, 128, (size_t)w);
        if (right) std::memset(right + y * stride, 128, (size_t)w);
    }
    const int nrect = (int)(((int64_t)w * h) / 200);
    for (int i = 0; i < nrect; ++i) {
        const int rw = 6 + (int)rng.below(43), rh = 6 + (int)rng.below(43);
        const int x0 = (int)rng.below((uint32_t)w), y0 = (int)rng.below((uint32_t)h);
        const uint8_t v = (uint8_t)(16 + rng.below(225));
        const int disp = (int)rng.below(41);
        paint(left, w, h, stride, x0, y0, rw, rh, v);
        if (right) paint(right, w, h, stride, x0 - disp, y0, rw, rh, v);
    }
    for (int pass = 0; pass < (right ? 2 : 1); ++pass) {
        uint8_t* img = pass ? right : left;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int v = img[y * stride + x] + (int)rng.below(13) - 6;
                img[y * stride + x] = (uint8_t)std::min(std::max(v, 0), 255);
            }
    }
}

} // namespace orbx

using namespace orbx;

extern "C" {

const char* orbx_status_string(int s)
{
    switch (s) {
    case ORBX_OK: return "ok";
    case ORBX_ERR_BAD_ARG: return "bad argument";
    case ORBX_ERR_UNSUPPORTED_GEOMETRY: return "unsupported image geometry";
    case ORBX_ERR_HIP: return "HIP runtime error";
    case ORBX_ERR_CAPACITY: return "output capacity too small";
    case ORBX_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
    }
}

const char* orbx_last_error(void) { return g_last_error.c_str(); }

#ifndef ORBX_BUILD_STAMP
#define ORBX_BUILD_STAMP "unstamped"
#endif
const char* orbx_build_stamp(void) { return ORBX_BUILD_STAMP; }

// Contiguous split of `total` units (frames of a batch, rows of a descriptor map) over `world` ranks: rank r owns
// [r*total/world, (r+1)*total/world), so the ranges tile [0, total) in rank order and differ in size by at most one.
int orbx_shard_range(int64_t total, int rank, int world, int64_t* begin, int64_t* end)
{
    if (total < 0 || world < 1 || rank < 0 || rank >= world || !begin || !end) { set_error("orbx_shard_range: bad argument"); return ORBX_ERR_BAD_ARG; }
    *begin = (int64_t)(((__int128)total * rank) / world);
    *end = (int64_t)(((__int128)total * (rank + 1)) / world);
    return ORBX_OK;
}

int orbx_scale_tables(const orbx_params* p, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                      int32_t* features_per_level, int32_t* umax16)
{
    if (!p) return ORBX_ERR_BAD_ARG;
    ScaleTables t;
    int rc = build_scale_tables(*p, t);
    if (rc) return rc;
    for (int i = 0; i < t.nlevels; ++i) {
        if (scale) scale[i] = t.scale[i];
        if (inv_scale) inv_scale[i] = t.inv_scale[i];
        if (sigma2) sigma2[i] = t.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = t.inv_sigma2[i];
        if (features_per_level) features_per_level[i] = t.quota[i];
    }
    if (umax16) for (int i = 0; i < 16; ++i) umax16[i] = t.umax[i];
    return ORBX_OK;
}

int orbx_level_size(const orbx_params* p, int width, int height, int level, int* w, int* h)
{
    if (!p || !w || !h) return ORBX_ERR_BAD_ARG;
    ScaleTables t;
    int rc = build_scale_tables(*p, t);
    if (rc) return rc;
    if (level < 0 || level >= t.nlevels) return ORBX_ERR_BAD_ARG;
    *w = cv_round((float)width * t.inv_scale[level]);
    *h = cv_round((float)height * t.inv_scale[level]);
    return ORBX_OK;
}

int orbx_keypoint_capacity(const orbx_params* p, int width, int height)
{
    if (!p) return -ORBX_ERR_BAD_ARG;
    Geometry g;
    int rc = build_geometry(*p, width, height, g);
    if (rc) return -rc;
    return g.out_cap;
}

// ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1650-1666
int orbx_descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t x, y;
        std::memcpy(&x, a + 8 * i, 8);
        std::memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

// ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1604-1645
void orbx_compute_three_maxima(const int32_t* sizes, int L, int* ind1, int* ind2, int* ind3)
{
    int m1 = 0, m2 = 0, m3 = 0;
    for (int i = 0; i < L; ++i) {
        const int s = sizes[i];
        if (s > m1) { m3 = m2; m2 = m1; m1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > m2) { m3 = m2; m2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > m3) { m3 = s; *ind3 = i; }
    }
    if (m2 < 0.1f * (float)m1) { *ind2 = -1; *ind3 = -1; }
    else if (m3 < 0.1f * (float)m1) { *ind3 = -1; }
}

// Merge of per-shard best/second records: lowest distance, then lowest global index; the second is
// the second smallest of the union, which is what one sequential pass over the whole set yields.
int orbx_best2_merge(const orbx_best2* parts, int n_parts, int nq, orbx_best2* out)
{
    if (!parts || !out || n_parts < 1 || nq < 0) return ORBX_ERR_BAD_ARG;
    for (int qi = 0; qi < nq; ++qi) {
        int best = 256, second = 256, idx = -1;
        for (int p = 0; p < n_parts; ++p) {
            const orbx_best2& r = parts[(size_t)p * nq + qi];
            if (r.best_idx < 0) continue;
            if (r.best < best || (r.best == best && r.best_idx < idx)) { second = std::min(second, best); best = r.best; idx = r.best_idx; }
            else second = std::min(second, r.best);
            second = std::min(second, r.second);
        }
        out[qi].best = best; out[qi].best_idx = idx; out[qi].second = second; out[qi].pad = 0;
    }
    return ORBX_OK;
}

struct orbx_frame_grid { FrameGrid g; };

int orbx_grid_create(const orbx_keypoint* kps, int n, float minX, float maxX, float minY, float maxY, orbx_frame_grid** out)
{
    if (!out || n < 0 || (!kps && n > 0) || !(maxX > minX) || !(maxY > minY)) return ORBX_ERR_BAD_ARG;
    orbx_frame_grid* g = new orbx_frame_grid();
    g->g.build(kps, n, minX, maxX, minY, maxY);
    *out = g;
    return ORBX_OK;
}
void orbx_grid_destroy(orbx_frame_grid* g) { delete g; }

int orbx_grid_features_in_area(const orbx_frame_grid* g, float x, float y, float r, int minLevel, int maxLevel,
                               int32_t* out, int cap, int* n)
{
    if (!g || !n) return ORBX_ERR_BAD_ARG;
    std::vector<int> v;
    g->g.query(x, y, r, minLevel, maxLevel, v);
    *n = (int)v.size();
    if ((int)v.size() > cap) return ORBX_ERR_CAPACITY;
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return ORBX_OK;
}

void orbx_synth_frame(uint32_t frame_index, int w, int h, uint8_t* out, ptrdiff_t stride)
{
    synth(frame_index, w, h, out, nullptr, stride);
}
void orbx_synth_stereo(uint32_t frame_index, int w, int h, uint8_t* left, uint8_t* right, ptrdiff_t stride)
{
    synth(frame_index, w, h, left, right, stride);
}
void orbx_synth_descriptors(uint32_t seed, int64_t n, uint8_t* out)
{
    Pcg32 rng(seed);
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    for (int64_t i = 0; i < n * 8; ++i) { uint32_t v = rng.next(); std::memcpy(o + i, &v, 4); }
}
void orbx_synth_queries(uint32_t seed, const uint8_t* map, int64_t n_map, int nq, uint8_t* out, int64_t* src_idx)
{
    Pcg32 rng(seed);
    for (int i = 0; i < nq; ++i) {
        const int64_t r = (int64_t)((((uint64_t)rng.next() << 32) | rng.next()) % (uint64_t)n_map);
        if (src_idx) src_idx[i] = r;
        std::memcpy(out + (size_t)i * 32, map + (size_t)r * 32, 32);
        for (int b = 0; b < 256; ++b)
            if (rng.below(100) < 8) out[(size_t)i * 32 + (b >> 3)] ^= (uint8_t)(1u << (b & 7));
    }
}

int64_t orbx_mat_record_bytes(int rows, int cols, int elem_size)
{
    if (rows < 0 || cols < 0 || elem_size < 1) return -1;
    return 24 + (int64_t)rows * cols * elem_size;
}

int64_t orbx_mat_record_write(const uint8_t* data, int rows, int cols, int elem_size, int elem_type, uint8_t* out, int64_t cap)
{
    const int64_t need = orbx_mat_record_bytes(rows, cols, elem_size);
    if (need < 0 || !out || cap < need || (need > 24 && !data)) return -1;
    const int32_t c = cols, r = rows;
    const uint64_t es = (uint64_t)elem_size, et = (uint64_t)elem_type;          // size_t of the LP64 builds the reference targets
    std::memcpy(out, &c, 4); std::memcpy(out + 4, &r, 4); std::memcpy(out + 8, &es, 8); std::memcpy(out + 16, &et, 8);
    if (need > 24) std::memcpy(out + 24, data, (size_t)(need - 24));
    return need;
}

int64_t orbx_mat_record_read(const uint8_t* rec, int64_t len, int* rows, int* cols, int* elem_size, int* elem_type, const uint8_t** data)
{
    if (!rec || len < 24) return -1;
    int32_t c, r; uint64_t es, et;
    std::memcpy(&c, rec, 4); std::memcpy(&r, rec + 4, 4); std::memcpy(&es, rec + 8, 8); std::memcpy(&et, rec + 16, 8);
    if (c < 0 || r < 0 || es < 1 || es > 64 || et > 0x7FFFFFFFu) return -1;
    const int64_t need = 24 + (int64_t)r * c * (int64_t)es;
    if (len < need) return -1;
    if (rows) *rows = r;
    if (cols) *cols = c;
    if (elem_size) *elem_size = (int)es;
    if (elem_type) *elem_type = (int)et;
    if (data) *data = rec + 24;
    return need;
}

int orbx_synth_vocabulary(uint32_t seed, int k, int L, int32_t* parent, uint8_t* is_leaf, uint8_t* descriptors, double* weight)
{
    if (k < 2 || k > 20 || L < 1 || L > 10) return -1;
    long long n = 1, level_nodes = 1;
    for (int l = 1; l <= L; ++l) { level_nodes *= k; n += level_nodes; if (n > 0x7FFFFFFF) return -1; }
    if (!parent) return (int)n;                                  // size query
    Pcg32 rng(seed);
    // breadth-first numbering: the children of node p are nodes p*k+1 .. p*k+k; a child's descriptor is its parent's
    // with a share of bits flipped that shrinks with depth, so the descent has a meaningful nearest child
    parent[0] = -1; is_leaf[0] = 0; weight[0] = 0; std::memset(descriptors, 0, 32);
    long long first_of_level = 1; level_nodes = 1;
    int words = 0;
    for (int l = 1; l <= L; ++l) {
        level_nodes *= k;
        for (long long i = first_of_level; i < first_of_level + level_nodes; ++i) {
            const long long p = (i - 1) / k;
            parent[i] = (int32_t)p;
            is_leaf[i] = l == L;
            weight[i] = l == L ? 0.5 + (double)(words++ % 7) : 0.0;
            for (int wd = 0; wd < 8; ++wd) {
                uint32_t v;
                if (l == 1) v = rng.next();
                else {
                    uint32_t flip = rng.next() & rng.next() & rng.next();
                    for (int e = 2; e < std::min(l, 4); ++e) flip &= rng.next();
                    std::memcpy(&v, descriptors + (size_t)p * 32 + wd * 4, 4);
                    v ^= flip;
                }
                std::memcpy(descriptors + (size_t)i * 32 + wd * 4, &v, 4);
            }
        }
        first_of_level += level_nodes;
    }
    return (int)n;
}

} // extern "C"
