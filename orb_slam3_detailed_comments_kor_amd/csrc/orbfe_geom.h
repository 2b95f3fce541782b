/*
 * orbfe_geom.h -- per-level / per-cell geometry shared by the host shim and the kernels.
 *
 * Everything here is derived on the host exactly as the reference derives it
 * (ORBextractor ctor src/ORBextractor.cc:408-468, ComputePyramid :1152-1177,
 * the cell grid of ComputeKeyPointsOctTree :769-804, the roots of DistributeOctTree
 * :540-556) and uploaded once per image size.
 */
#ifndef ORBFE_GEOM_H
#define ORBFE_GEOM_H

#include <stdint.h>

#define ORBFE_MAX_LEVELS 16
#define ORBFE_EDGE 19          /* EDGE_THRESHOLD, src/ORBextractor.cc:72  */
#define ORBFE_MINB 16          /* EDGE_THRESHOLD-3, :771                  */
#define ORBFE_ROI_X0 64        /* byte column of ROI x=0 in a pyramid row (64-B aligned rows) */
#define ORBFE_MAX_DIM 4096     /* packed candidate = x | y<<12 | score<<24 */
#define ORBFE_FAST_TILE 76     /* max ROI side: wCell+6 <= 75              */

struct OrbLevelGeom {
    int w, h;             /* level size (cvRound((float)cols*inv), :1157)                     */
    int pitch;            /* row pitch of the padded level buffer                            */
    uint32_t bufOff;      /* byte offset of the padded buffer inside one image's pyramid slab */
    uint32_t roiOff;      /* byte offset of ROI(0,0) = bufOff + 19*pitch + ORBFE_ROI_X0       */
    int nCols, nRows, wCell, hCell;
    int cellBase, nCells; /* range in the cell table                                          */
    int maxBX, maxBY;     /* w-16, h-16                                                       */
    int nFeat;            /* mnFeaturesPerLevel[level]                                        */
    int nIni;             /* quadtree roots, :540                                             */
    float hX;             /* :542                                                             */
    int kpBase, kpCap;    /* slot range of this level's quadtree output                       */
    int keyBase, keyCap;  /* slot range of this level's compacted candidate list              */
    int listCap;          /* quadtree node-list capacity                                      */
    float scale;          /* mvScaleFactor[level]                                             */
    float size;           /* (float)(int)(31*scale), :862                                     */
    int xtabOff, ytabOff; /* resize tables of this level (level > 0)                          */
};

struct OrbCellGeom {
    int16_t level;
    int16_t iniX, iniY; /* ROI origin in level coordinates        */
    int16_t cw, ch;     /* ROI size (maxX-iniX, maxY-iniY)        */
    int16_t offX, offY; /* j*wCell, i*hCell (added to keypoints)  */
    int16_t nd;         /* dwords per staged tile row: (cw + (iniX&3) + 3) / 4 */
    int32_t slotBase;   /* first candidate slot of this cell      */
    int32_t slotCap;    /* ceil(zw/2)*ceil(zh/2): max strict 8-neighbour local maxima */
    /* ceil(2^32/d) reciprocals (0 encodes d == 1): x/d == umulhi(x, m) while x*d < 2^32 */
    uint32_t mNd, mNdz, mZw;
    uint32_t roiOff;    /* copy of the level's roiOff and pitch: saves the kernel a dependent load */
    int32_t pitch;
    uint32_t mZh;       /* reciprocal of the zone height ch - 6 */
    int32_t pad2[2];
};

/* K-FAST's own view of a cell: 32 B, fetched with one scalar load.  Everything the kernel would otherwise derive per
 * workgroup is folded on the host (tile origin as one byte offset, zone geometry, reciprocal). */
struct OrbFastCell {
    uint32_t gOff;     /* byte offset of the tile origin (ROI row iniY, column iniX & ~3) inside one image's pyramid slab */
    uint32_t pitch;    /* row pitch of the level                                                                        */
    uint32_t dims;     /* cw | ch << 8 | (iniX & 3) << 16 | ndz << 20  (ndz = dword columns of the detection zone)      */
    uint32_t off;      /* offX | offY << 16 (added to the candidates' coordinates)                                      */
    uint32_t slotBase; /* first candidate slot of this cell                                                             */
    uint32_t slotCap;
    uint32_t mNdz;     /* ceil(2^32 / ndz), 0 encodes ndz == 1                                                          */
    uint32_t pad;
};

/* K-FAST, round 4: a workgroup's unit of work is a RUN of up to ORBFE_FAST_RUN_MAXC cells of one cell row (one coalesced tile,
 * the aprons between the cells shared, one survivor stream for the score phase, cell seams as masks in the NMS, one rank
 * space per cell).  48 B, host-built per image size.  The cells of a run are consecutive in the cell table and in the
 * candidate slots; all but the last have the level's wCell as their zone width. */
#define ORBFE_FAST_RUN_MAXC 4
struct OrbFastRun {
    uint32_t gOff;     /* byte offset of the tile origin (ROI row iniY, column iniX & ~3 of the first cell) in an image's slab */
    uint32_t pitch;    /* row pitch of the level                                                                            */
    uint32_t dims;     /* tile width in px (ox + zone + 6) | rows << 10 | ox << 18 | cells << 20                             */
    uint32_t off;      /* offX | offY << 16 of the first cell (added to the candidates' coordinates)                         */
    uint32_t cell0;    /* index of the first cell in the cell table (cellCount entry)                                       */
    uint32_t slotBase; /* first candidate slot of the first cell                                                            */
    uint32_t zw;       /* zone width of every cell but the last | of the last << 16                                         */
    uint32_t cap;      /* slot capacity of every cell but the last | of the last << 16                                      */
    uint32_t mZw;      /* ceil(2^32 / zone width of the non-last cells)                                                     */
    uint32_t mNdz;     /* ceil(2^32 / dword columns of the run's zone)                                                      */
    uint32_t ndz;      /* dword columns of the run's zone                                                                   */
    uint32_t pad;
};

/* resize tables: per destination column / row */
struct OrbResizeX {
    uint16_t sx;
    int16_t a0, a1;
    uint16_t pad;
};
struct OrbResizeY {
    uint16_t sy0, sy1; /* clamped source rows */
    int16_t b0, b1;
};

/* Fused pyramid kernel: per (level, tile index) ranges along one axis.  Tile i owns [lo, ownHi) of the
 * level (the ranges of all tiles partition the level) and must compute [lo, needHi) so that the next
 * level's needed range can be interpolated from it (halo, recomputed identically by neighbours). */
struct OrbPyrRange {
    int16_t lo, ownHi, needHi, pad;
};
#define ORBFE_PYR_TILE 24 /* tile side at the coarsest level (16: 157 us, 24: 130 us, 32: 129 us per 64 frames) */

/* Everything a workgroup of the fused pyramid kernel needs about ITS tile, built on the host once per image size and
 * copied into LDS as one block (round 3; the kernel used to derive it per workgroup: range lookups, table staging with a
 * level search per entry, selector arithmetic).  Record of tile (ti, tj), at (tj * ntx + ti) * recBytes:
 *   OrbPyrTileHdr | xsel[stageX] (uint4: the four v_perm_b32 selectors of a group of four destination columns)
 *                 | xa[stageX] (uint4: a0 | a1 << 16 of the four columns) | xb[align4(stageX)] (u32: dword-aligned source
 *                   column of the group | byte shift << 16) | yt[align2(stageY)] (uint2: LDS byte offsets of the two source
 *                   rows sy0 | sy1 << 16 inside the source region, b0 | b1 << 16)
 * stageX / stageY = the largest group / row totals of any tile (the kernel's pointer offsets). */
struct OrbPyrTileHdr {
    int32_t xlo[ORBFE_MAX_LEVELS], xown[ORBFE_MAX_LEVELS], xneed[ORBFE_MAX_LEVELS]; /* owned [lo, own), computed [lo, need) */
    int32_t ylo[ORBFE_MAX_LEVELS], yown[ORBFE_MAX_LEVELS], yneed[ORBFE_MAX_LEVELS];
    int32_t roi[ORBFE_MAX_LEVELS], pitch[ORBFE_MAX_LEVELS];                         /* level geometry                   */
    uint32_t recip[ORBFE_MAX_LEVELS]; /* ceil(2^32 / column groups per region row), 0 when there is one */
    int32_t xo[ORBFE_MAX_LEVELS + 1], yo[ORBFE_MAX_LEVELS + 1]; /* first staged x group / y entry of a level; [nlevels] = total */
    int32_t pad[2];
};

/* K-DESC's view of one slot of an image's level-major keypoint slot array (slot g = lg[level].kpBase + k, the k-th
 * keypoint K-QT kept at that level): 32 B per slot, the same for every image of a size, fetched with one scalar load.
 * With it a wavefront needs nothing from K-PACK: its level's geometry is here, its key is lvlKp[g], its output slot the
 * number of keypoints of the lower levels plus k. */
struct OrbDescSlot {
    uint32_t roiOff; /* level geometry: byte offset of ROI(0,0) inside one image's pyramid slab */
    int32_t pitch;
    uint32_t wh;     /* w | h << 16                                                              */
    uint32_t lk;     /* level | k << 8                                                           */
    float scale;     /* mvScaleFactor[level]                                                     */
    float size;      /* keypoint size of the level                                               */
    int32_t pad[2];
};

#endif
