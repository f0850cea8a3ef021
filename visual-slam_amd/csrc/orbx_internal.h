// Internal declarations shared by the host-only and the HIP translation units of liborbx.
#pragma once
#include "../../include/orbx.h"
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>

namespace orbx {

constexpr int kEdge = 19;        // EDGE_THRESHOLD, src/ORBextractor.cc:73
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE, src/ORBextractor.cc:72
constexpr int kPatch = 31;       // PATCH_SIZE, src/ORBextractor.cc:71
constexpr int kMinBorder = 16;   // EDGE_THRESHOLD-3, src/ORBextractor.cc:772
constexpr int kMaxLevels = 32;
constexpr int kRoiX = 32;        // column of the ROI origin inside a padded row (keeps ROI rows 32-byte aligned)
constexpr int kMaxRoots = 64;
constexpr int kFrameFill = 3;    // pixels of the REFLECT_101 frame that later stages read (7x7 Gaussian of ROI pixels)

void set_error(const std::string& s);

// Raises the dynamic-LDS limit of kernel fn on the current device to at least `bytes` (a hipError_t as int).  The limit
// belongs to the kernel and the device, not to a launch or a context: it only ever goes up, so that contexts with
// different needs (a 3840x2160 extractor beside a 640x480 one) do not lower it under each other.  Defined in orbx_extract.hip.
int raise_dynamic_lds(const void* fn, size_t bytes);

// Constructor tables of ORBextractor (src/ORBextractor.cc:409-469)
struct ScaleTables {
    int nlevels = 0;
    float scale[kMaxLevels], inv_scale[kMaxLevels], sigma2[kMaxLevels], inv_sigma2[kMaxLevels];
    int quota[kMaxLevels];
    int umax[16];
};
int build_scale_tables(const orbx_params& p, ScaleTables& t);

// One FAST cell of ComputeKeyPointsOctTree (src/ORBextractor.cc:788-828), in ROI coordinates.
struct CellRec {
    int16_t x0, y0;      // top-left of the cell sub-image (iniX, iniY)
    int16_t cw, ch;      // size of the detection area (sub-image size minus the 3-pixel ring on each side)
    int16_t offx, offy;  // j*wCell, i*hCell: added to cell-local coordinates (:821-822)
    int16_t level, pad;
    int32_t slot;        // first candidate slot of this cell inside its level's slot array
    int32_t cap;         // slots reserved: ceil(cw/2)*ceil(ch/2), the most strict local maxima a cw x ch area can hold
};

struct ResizeX { int32_t sx; int16_t a0, a1; };          // source column and Q11 weights of one destination column
// Four adjacent destination columns (a group): first source column, byte offset of each column's left
// neighbour from it (one byte each), and the packed Q11 weights a0 | a1 << 16 of each column, split into a
// 16-byte and an 8-byte array so that consecutive lanes read consecutive words.  Group 0 of a level is the
// frame dword left of the ROI (columns -4..-1), group 1 + x/4 holds ROI column x.
struct ResizeGroupA { int32_t sx_first; uint32_t offs; uint32_t w01[2]; };
struct ResizeGroupB { uint32_t w23[2]; };
struct ResizeY { int16_t sy0, sy1; int16_t b0, b1; };    // clamped source rows and Q11 weights of one destination row
// Four consecutive destination rows whose source rows are base+r or base+r+1 (and the row below, clamped at the last row):
// w[r] = b0 | b1 << 12 | (sy0 == base+r+1) << 24; bits 25..31 of w[0], w[1], w[2] hold base (7 bits each, low first)
struct ResizeRowBlock { uint32_t w[4]; };

struct LevelGeom {
    int w, h;               // ROI size (src/ORBextractor.cc:1111)
    int stride, rows;       // padded buffer: rows = h+38, stride bytes per row
    size_t offset;          // byte offset of the padded buffer inside one frame's pyramid block
    size_t roi;             // byte offset of the ROI origin inside one frame's pyramid block
    int maxBorderX, maxBorderY, nCols, nRows, wCell, hCell;
    int cell_first, cell_count;
    int cand_offset, cand_cap;   // candidate slots of this level inside one frame's slot array (u32 units)
    int quota;              // mnFeaturesPerLevel[level]
    int nIni; float hX;     // DistributeOctTree roots (src/ORBextractor.cc:542-544)
    int kp_offset, kp_cap;  // selected-keypoint slots of this level inside one frame's per-level array
    int area2x;             // resize from the previous level is an exact 2x2 decimation (OpenCV reroutes to INTER_AREA)
    int rx_offset, ry_offset; // offsets into the ResizeX / ResizeY tables (rx_offset is a multiple of 4)
    int rb_offset;            // first ResizeRowBlock of this level (fast_resize == 2)
    int fast_resize;        // 1: every aligned group of 4 destination columns reads <= 8 consecutive source bytes; 2: and every block of 4 rows fits ResizeRowBlock
    int rg_offset;          // first group record of this level
    int gpr;                // groups per row that the resize kernel writes: 1 + ceil((w + kFrameFill) / 4)
    unsigned gpr_magic;     // ceil(2^32 / gpr)
    int cpr;                // 16-byte chunks per row that the pyramid kernels write, ceil((w + kFrameFill)/16)
    unsigned cpr_magic;     // ceil(2^32 / cpr)
    int tpr;                // 16-byte columns per row of the tiled blurred level, ceil(w/16) + 2
    int upr;                // 16-pixel units per row that the blur kernel produces, ceil(w/16)
    unsigned upr_magic;     // ceil(2^32 / upr)
    size_t blur_offset;     // byte offset of the tiled blurred level inside one frame's blurred block
    float scale;            // mvScaleFactor[level]
    float kp_size;          // (float)(int)(31*scale), src/ORBextractor.cc:836,845
};

struct Geometry {
    int width = 0, height = 0, nlevels = 0;
    ScaleTables tables;
    std::vector<LevelGeom> lv;
    std::vector<CellRec> cells;
    std::vector<ResizeX> rx;
    std::vector<ResizeY> ry;
    std::vector<ResizeRowBlock> rrb;
    std::vector<ResizeGroupA> rga;
    std::vector<ResizeGroupB> rgb;
    size_t pyr_bytes = 0;     // one frame's pyramid block
    size_t blur_bytes = 0;    // one frame's blurred block (every level tiled: 8 rows x 16 bytes per 128-byte tile)
    int cand_total = 0;       // one frame's candidate slots (u32)
    int kp_total = 0;         // one frame's per-level selected-keypoint slots
    int out_cap = 0;          // sum over levels of kp_cap
    int max_cell_dim = 0;
};
// returns ORBX_OK or ORBX_ERR_UNSUPPORTED_GEOMETRY / ORBX_ERR_BAD_ARG
int build_geometry(const orbx_params& p, int width, int height, Geometry& g);

// 64x48 frame grid (src/Frame.cc:230-245, 327-392)
struct FrameGrid {
    const orbx_keypoint* kps; int n;
    float minX, maxX, minY, maxY, invW, invH;
    std::vector<int> cell_start, cell_items;
    void build(const orbx_keypoint* k, int n_, float minX_, float maxX_, float minY_, float maxY_);
    void query(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const;
};

// device CSR distance kernel driver (defined in orbx_match.hip); host pointers in/out
int csr_distances(orbx_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt,
                  const int32_t* rowptr, const int32_t* cand, int32_t* dist_out);

// One windowed search: Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (src/Frame.cc:327-380); r < 0 asks for nothing.
struct WindowQuery { float x, y, r; int16_t minLevel, maxLevel; };
// Candidate lists and distances of nq windowed searches, on the device: row i of the CSR holds, in the order the
// reference's grid walk visits them (grid_order = the frame grid's cell_items: cells column by column, insertion order
// inside a cell), the keypoints of k2 inside query i's window and level band, with the Hamming distance between
// qdesc[i] and their descriptor.  Defined in orbx_match.hip.
int window_csr(orbx_matcher* m, const WindowQuery* q, const uint8_t* qdesc, int nq,
               const orbx_keypoint* k2, const uint8_t* d2, int n2, const int* grid_order, int n_in_grid,
               std::vector<int32_t>& rowptr, std::vector<int32_t>& cand, std::vector<int32_t>& dist);

// device, private stream and growable device staging buffers (slot 0..7) of a matcher context (orbx_match.hip)
int matcher_device(const orbx_matcher* m);
void* matcher_stream(const orbx_matcher* m);
int matcher_stage(orbx_matcher* m, int slot, size_t bytes, void** out);

} // namespace orbx
