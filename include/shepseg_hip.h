/*
 * shepseg_hip.h -- C-ABI of libshepseg_hip.so, the MI355X (gfx950) implementation of the
 * pyshepseg per-tile segmentation hot path and its tiled driver.
 *
 * The reference (ubarsc/pyshepseg v2.0.3) has no native/FFI boundary: its operator API is
 * three Python call signatures whose arithmetic runs in numba @njit functions and in
 * sklearn.cluster.KMeans.  Each entry point below replaces the reference function(s)
 * cited next to it; pyshepseg_amd/{shepseg,tiling,tilingstats}.py bind them with ctypes
 * under the reference's own Python names (see INTEGRATION.md for the stub).
 *
 * Conventions
 *  - return 0 = OK, negative = error; message via shp_last_error(ctx).
 *  - the caller owns every host buffer; the library borrows pointers for the call only.
 *  - all arrays C-contiguous; images are band-planar (nBands, nRows, nCols) like the
 *    reference's `img` (shepseg.py:140).
 *  - one shp_ctx per host thread / HIP stream; calls on different contexts are re-entrant
 *    (the reference calls doShepherdSegmentation from N threads, tiling.py:1560-1595).
 *  - there is NO CPU fallback: every entry point fails with SHP_ERR_NO_DEVICE when no
 *    gfx950 device is usable.
 */
#ifndef SHEPSEG_HIP_H
#define SHEPSEG_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct shp_ctx shp_ctx;

/* image dtypes (the integer dtypes the reference accepts for `img`) */
enum { SHP_U8 = 0, SHP_I16 = 1, SHP_U16 = 2, SHP_I32 = 3, SHP_U32 = 4 };

enum {
    SHP_OK = 0,
    SHP_ERR_NO_DEVICE = -1,
    SHP_ERR_HIP = -2,
    SHP_ERR_ARG = -3,
    SHP_ERR_NOMEM = -4,
    SHP_ERR_STATE = -5
};

int shp_version(void);
int shp_device_count(void);                      /* number of usable HIP devices (0 = none) */
int shp_ctx_create(int device, shp_ctx **out);   /* owns one HIP stream + device workspace */
int shp_ctx_create_priority(int device, shp_ctx **out);  /* same, highest stream priority */
/* A worker context of the tiled drivers: device workspace without a stream of its own.  Inside the
 * worker calls (shp_assign_rects_dev, shp_segment_window_dev, shp_segment_tile_to_dev,
 * shp_stitch_prepare_dev) it borrows a stream of a process-wide pool for each phase (SHEPSEG_FILL_MAX
 * fill streams, SHEPSEG_WALK_STREAMS = 12 walker streams), so that more tiles than hardware queues
 * can be in flight; other calls run on one idle stream all shared contexts use. */
int shp_ctx_create_shared(int device, shp_ctx **out);
/* Grow the context's (grow-only) workspace for tiles of up to npix pixels of this type now, instead
 * of in the middle of a run when the first such tile arrives.  The reference has no counterpart
 * (numpy allocates per call); the tiled drivers call it once per worker thread
 * (SegThreadsMgr.worker, tiling.py:1560-1600) with the job's largest tile. */
int shp_ctx_reserve(shp_ctx *ctx, int dtype, int nbands, int64_t npix);
/* what that reservation would still allocate on this context (bytes), and the device's free and
 * total memory: the tiled driver starts only as many workers as fit (SegThreadsMgr's numWorkers,
 * tiling.py:1531-1600, has no such limit because its tiles live in host memory) */
int shp_ctx_reserve_query(shp_ctx *ctx, int dtype, int nbands, int64_t npix, int64_t *extra_bytes,
                          int64_t *free_bytes, int64_t *total_bytes);
void shp_ctx_destroy(shp_ctx *ctx);
const char *shp_last_error(const shp_ctx *ctx);  /* valid until the next call on ctx */
/* device-time (HIP events on the ctx stream) of the stages of the last shp_segment_tile /
 * stage call, milliseconds: [0]=assign [1]=clump [2]=single-pixel [3]=small-segment
 * [4]=h2d [5]=d2h [6]=total.  out must hold 8 doubles. */
int shp_last_timings(const shp_ctx *ctx, double *out);
/* accumulated device time (ms, HIP events on the ctx stream) and launch count of the
 * instrumented kernels: ids 0 assign, 1 ccl, 2 dfs_pool (the replay), 3 radix sort, 4 spectra,
 * 5 small-segment pass loop, 7 seed scan + final labels.  reset != 0 clears the counters. */
int shp_prof_get(shp_ctx *ctx, double *ms_out, uint64_t *count_out, int n, int reset);

/* ---- k-means ------------------------------------------------------------------------ */
/* replaces sklearn KMeans(init=<array>, n_init=1).fit as called by
 * shepseg.fitSpectralClusters (shepseg.py:305-312).  xsample: nrows*nbands float64 rows.
 * The reference's sklearn 0.24.2 runs ELKAN's variant for k > 1 (algorithm="auto").  The fit runs
 * Lloyd iterations (same partitions, far fewer bytes) under a guard and starts over with Elkan's
 * algorithm as the reference evaluates it -- bounds, strict-improvement relabelling, row-order sums --
 * as soon as a sample's two nearest centres are within 1e-12 (relative): there the result is the
 * reference's with one OpenMP thread bit for bit (pyshepseg_amd/csrc/fit_elkan.h).
 * SHEPSEG_FIT_ALGO=lloyd|elkan forces one path. */
int shp_kmeans_fit(shp_ctx *ctx, const double *xsample, int64_t nrows, int nbands, int k,
                   const double *init_centres, int max_iter, double tol_rel,
                   double *centres_out, int32_t *labels_out, int *n_iter_out);
/* the same with the sample rows in the image's pixel type (SHP_U8 ... SHP_U32): sklearn's
 * check_array conversion to float64 happens on the fly, the arithmetic is unchanged. */
int shp_kmeans_fit_typed(shp_ctx *ctx, const void *xsample, int dtype, int64_t nrows, int nbands,
                         int k, const double *init_centres, int max_iter, double tol_rel,
                         double *centres_out, int32_t *labels_out, int *n_iter_out);
/* The same from the BAND-PLANAR sub-sample (nbands planes of npix pixels, as shp_dev_subsample
 * returns it): rows with null_val in any band are dropped when has_null, init_centres == NULL means
 * the reference's diagonalClusterCentres (shepseg.py:364-397) of the rows kept; *nrows_out = rows the
 * model was fitted on (labels_out, optional, receives that many).  Identical arithmetic (each band's
 * sums are one chain in row order), prepared by one host thread per band, no host transposition. */
int shp_kmeans_fit_planar(shp_ctx *ctx, const void *planes, int dtype, int64_t npix, int nbands,
                          int has_null, int64_t null_val, int k, const double *init_centres,
                          int max_iter, double tol_rel, double *centres_out, int32_t *labels_out,
                          int *n_iter_out, int64_t *nrows_out);

/* The same with the E-step of the reference's algorithm SHARDED BY SAMPLE ROWS over the ranks of an RCCL
 * communicator (shp_comm_create): every rank passes the SAME sample and receives the SAME model.  Rank r keeps
 * the bounds of rows [r n/N, (r+1) n/N) and relabels them; the labels are all-gathered in place on the fit's
 * stream every iteration (ncclAllGather, no host round trip) and every rank runs the M-step on all of them --
 * same sums in the same order, so no broadcast of centres.  The reference's fit is one process
 * (shepseg.py:305-312); bit-identical to shp_kmeans_fit_planar.  cm == NULL or one rank: that call. */
struct shp_comm;
int shp_kmeans_fit_planar_dist(shp_ctx *ctx, struct shp_comm *cm, const void *planes, int dtype, int64_t npix,
                               int nbands, int has_null, int64_t null_val, int k, const double *init_centres,
                               int max_iter, double tol_rel, double *centres_out, int32_t *labels_out,
                               int *n_iter_out, int64_t *nrows_out);

/* which path the context's last fit took: 0 Lloyd iterations (no near tie met), 1 Elkan's */
int shp_last_fit_path(const shp_ctx *ctx);

/* replaces shepseg.applySpectralClusters (shepseg.py:317-361) + KMeans.predict:
 * clusters_out[nrows*ncols] int32, 1..k, 0 where any band == null_val. */
int shp_kmeans_assign(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows, int ncols,
                      const double *centres, int k, int has_null, int64_t null_val,
                      int32_t *clusters_out);

/* ---- per-tile stages (individually callable, like the reference's njit functions) ------ */
/* replaces shepseg.clump(img, ignoreVal=0, fourConnected, clumpId=1) (shepseg.py:452-541),
 * including the MAX_CLUMP_SIZE=10000 depth-first cut.  max_seg_id_out = next id - 1. */
int shp_clump(shp_ctx *ctx, const int32_t *clusters, int nrows, int ncols, int four_connected,
              uint32_t *seg_out, uint32_t *max_seg_id_out);

/* replaces shepseg.makeSegSize (shepseg.py:544-569): seg_size_out has max_seg_id+1 entries */
int shp_make_seg_size(shp_ctx *ctx, const uint32_t *seg, int64_t npix, uint32_t max_seg_id,
                      uint32_t *seg_size_out);

/* replaces shepseg.eliminateSinglePixels (shepseg.py:572-615): seg relabelled in place;
 * max_seg_id_inout: in = largest id in seg, out = seg.max() after the relabel. */
int shp_eliminate_single(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows,
                         int ncols, int four_connected, uint32_t *seg_inout,
                         uint32_t *max_seg_id_inout);

/* replaces shepseg.eliminateSmallSegments (shepseg.py:918-1000) */
int shp_eliminate_small(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows,
                        int ncols, int four_connected, int min_seg_size,
                        double max_spectral_diff, uint32_t *seg_inout,
                        uint32_t *max_seg_id_inout, int64_t *num_elim_out);

/* replaces shepseg.doShepherdSegmentation with a supplied k-means model
 * (shepseg.py:130-249, stages :206 :212 :219 :225 :235), fused on the device. */
int shp_segment_tile(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows, int ncols,
                     const double *centres, int k, int has_null, int64_t null_val,
                     int four_connected, int min_seg_size, double max_spectral_diff,
                     uint32_t *seg_out, uint32_t *max_seg_id_out, int64_t *singles_elim_out,
                     int64_t *small_elim_out, uint32_t *num_clumps_out);

/* ---- synthetic imagery (benchmark input; SURVEY.md Appendix B `synthimg v1`) ----------- */
int shp_synthimg(shp_ctx *ctx, uint64_t seed, int nbands, int64_t y0, int64_t x0, int nrows,
                 int ncols, uint16_t *out_host);

/* ---- device-resident rasters: tiled driver + cross-tile stitch --------------------------------
 * Device pointers cross the boundary as plain void* / uint32_t* (they come from shp_dev_alloc).
 * These replace the per-tile loop and the stitch of tiling.doTiledShepherdSegmentation:
 *   SegNoConcurrencyMgr.segmentAllTiles / SegThreadsMgr.worker (tiling.py:1413-1469, :1560-1600)
 *   stitchTiles / recodeTile / recodeSharedSegments / relabelSegments (tiling.py:950-1306)
 *   HistogramAccumulator (tiling.py:1915-1963), readSubsampledImageBand (tiling.py:259-314). */
int shp_dev_alloc(shp_ctx *ctx, size_t bytes, void **dptr);
int shp_dev_free(shp_ctx *ctx, void *dptr);
int shp_dev_upload(shp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int shp_dev_download(shp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int shp_dev_memset(shp_ctx *ctx, void *dst_dev, int value, size_t bytes);
int shp_dev_copy(shp_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);   /* D2D */
/* page-locked host buffers for the raster I/O pipeline (reads staged for H2D, finished output rows
 * staged from D2H): what the reference's per-tile GDAL ReadAsArray / WriteArray buffers become
 * (tiling.py:1436-1443, :1032-1034) */
int shp_host_alloc(shp_ctx *ctx, size_t bytes, void **hptr);
int shp_host_free(shp_ctx *ctx, void *hptr);
int shp_sync(shp_ctx *ctx);
/* synthimg v1 window written straight into device memory (band-planar uint16) */
int shp_dev_synthimg(shp_ctx *ctx, uint64_t seed, int nbands, int64_t y0, int64_t x0, int nrows,
                     int ncols, void *d_out);
/* synthetic label raster of block_rows x block_cols-pixel blocks numbered row-major from 1 (the
 * benchmark input of the statistics path, BASELINE config 5: "~50M segments"); *max_id_out = the
 * number of blocks.  Benchmark plumbing like shp_dev_synthimg: it replaces nothing in the reference. */
int shp_dev_block_labels(shp_ctx *ctx, int nrows, int ncols, int block_rows, int block_cols,
                         uint32_t *d_out, uint32_t *max_id_out);
/* out_host[b][i][j] = img[b][row_idx[i]][col_idx[j]] of a device raster (k-means subsample) */
int shp_dev_subsample(shp_ctx *ctx, const void *d_img, int dtype, int nbands, int nrows, int ncols,
                      const uint32_t *row_idx, int ny, const uint32_t *col_idx, int nx,
                      void *out_host);
/* doShepherdSegmentation on window (x, y, xs, ys) of a device raster; labels (ys*xs uint32,
 * local ids) are written to device memory d_seg_out.  Synchronous on return.  d_clusmap: NULL, or
 * the raster-wide cluster map whose window has been filled by shp_assign_rects_dev (the k-means
 * predict step of shepseg.py:211 is then not repeated for the pixels tiles share). */
int shp_segment_window_dev(shp_ctx *ctx, const void *d_img, int dtype, int nbands, int img_rows,
                           int img_cols, int x, int y, int xs, int ys, const double *centres, int k,
                           int has_null, int64_t null_val, int four_connected, int min_seg_size,
                           double max_spectral_diff, uint32_t *d_seg_out, uint32_t *max_seg_id_out,
                           int64_t *singles_elim_out, int64_t *small_elim_out,
                           uint32_t *num_clumps_out, const uint16_t *d_clusmap);
/* km.predict (shepseg.py:211, with the null mask of :205-207) on rectangles of a device raster:
 * d_clusmap[y][x] (uint16, img_rows x img_cols) = 0 for a null pixel, else cluster + 1.
 * rects = nrects x (x, y, xs, ys).  The model is global (tiling.py:154-226), so the tiled driver
 * assigns every pixel once instead of once per overlapping tile. */
int shp_assign_rects_dev(shp_ctx *ctx, const void *d_img, int dtype, int nbands, int img_rows,
                         int img_cols, const int32_t *rects, int nrects, const double *centres,
                         int k, int has_null, int64_t null_val, uint16_t *d_clusmap);
/* same with the tile image handed over as a host buffer (read -> H2D -> segment) */
int shp_segment_tile_to_dev(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows,
                            int ncols, const double *centres, int k, int has_null, int64_t null_val,
                            int four_connected, int min_seg_size, double max_spectral_diff,
                            uint32_t *d_seg_out, uint32_t *max_seg_id_out,
                            int64_t *singles_elim_out, int64_t *small_elim_out,
                            uint32_t *num_clumps_out);
/* one tile of stitchTiles, asynchronous on the ctx stream (call shp_sync to wait):
 * d_tile (ys*xs local ids) is recoded in place against the already-recoded strips of the tile
 * above (d_top_b: first row of its last `overlap` rows, row pitch top_pitch elements) and of the
 * tile to the left (d_left_b: first of its last `overlap` columns, pitch left_pitch); NULL where
 * there is no such neighbour.  The trimmed window [top,bottom) x [left,right) is written to
 * d_out at (yout, xout) (row pitch out_pitch) and *d_max_seg_id (device scalar) advances to the
 * largest id in it.  max_local = largest local id in d_tile. */
int shp_stitch_tile_dev(shp_ctx *ctx, uint32_t *d_tile, int ys, int xs, int overlap,
                        const uint32_t *d_top_b, int64_t top_pitch, const uint32_t *d_left_b,
                        int64_t left_pitch, uint32_t max_local, int simple_recode,
                        uint32_t *d_max_seg_id, int top, int bottom, int left, int right,
                        uint32_t *d_out, int64_t out_pitch, int xout, int yout);
/* The same stitch split in phases so that only a thin part is sequential (tiling.py:950-1306):
 *  shp_stitch_prepare_dev -- purely local to the tile, run by the worker that segmented it:
 *     fills d_meta = 4 x (max_local+1) uint32: flags (1 = crosses the top strip's midline,
 *     2 = crosses the left strip's, 4 = has a pixel in the trimmed window), bounding-box top row,
 *     bounding-box left column, and room for the LUT.  cross_px_out (may be NULL) receives the
 *     number of pixels of the top / left strip that belong to midline-crossing segments: handed
 *     to the chain call, it bounds the (segment, neighbour id) pair table there.  Synchronous.
 *  shp_stitch_chain_dev -- the sequential step (asynchronous on the ctx stream): modes over the
 *     overlap strips of the tile above / to the left (d_top_b / d_left_b as in
 *     shp_stitch_tile_dev, but they now point at the DENSE recoded strips written by earlier
 *     chain calls), new-id ranks, LUT, *d_max_seg_id advance, and the tile's own recoded right
 *     strip (ys x overlap, pitch overlap) / bottom strip (overlap x xs, pitch xs) into
 *     d_right_out / d_bottom_out (NULL = not needed).  The tile itself is not modified.  If
 *     d_out is not NULL the trimmed window is written through the LUT to d_out on the ctx's side
 *     stream, off the chain (shp_sync waits for both streams). */
int shp_stitch_prepare_dev(shp_ctx *ctx, const uint32_t *d_tile, int ys, int xs, int overlap,
                           int has_top, int has_left, uint32_t max_local, int top, int bottom,
                           int left, int right, uint32_t *d_meta, uint32_t *cross_px_out);
int shp_stitch_chain_dev(shp_ctx *ctx, const uint32_t *d_tile, int ys, int xs, int overlap,
                         const uint32_t *d_top_b, int64_t top_pitch, const uint32_t *d_left_b,
                         int64_t left_pitch, uint32_t max_local, int simple_recode,
                         uint32_t *d_max_seg_id, int top, int bottom, int left, int right,
                         uint32_t *d_meta, uint32_t *d_right_out, uint32_t *d_bottom_out,
                         uint32_t *d_out, int64_t out_pitch, int xout, int yout,
                         uint32_t top_cross_px, uint32_t left_cross_px /* 0xFFFFFFFF = unknown */);

/* Parallel stitch of the sharded driver: every tile runs the chain step with a
 * PROVISIONAL base (tile index * stride, in *d_max_seg_id) so that it depends on its two
 * neighbours' strips only, not on the running maxSegId of all earlier tiles (reference
 * tiling.py:1029-1043).  shp_stitch_counts_dev reports what the step did -- d_out2[0] = new ids
 * handed out, d_out2[1] = the largest of them present in the trimmed window, both relative to
 * base (asynchronous) -- and once every tile's count is known shp_renumber_dev maps
 * id -> new_base[id / stride] + id % stride over a raster (new_base: ntiles host values;
 * synchronous).  The result equals the sequential chain iff out2[0] == out2[1] for every tile;
 * otherwise the driver reruns the sequential chain. */
int shp_stitch_counts_dev(shp_ctx *ctx, const uint32_t *d_meta, uint32_t max_local, uint32_t base,
                          uint32_t *d_out2);
int shp_renumber_dev(shp_ctx *ctx, uint32_t *d_raster, int64_t npix, uint32_t stride,
                     const uint32_t *new_base, int ntiles);
/* one stitched, trimmed tile (w x h at xout, yout of the device raster) sub-sampled into one overview
 * layer exactly as SegmentationConcurrencyMgr.writeOverviews does tile by tile (tiling.py:1360-1383):
 * every level-th pixel from offset level / 2 of the tile, written at (xout / level, yout / level),
 * clipped to the ov_w x ov_h layer.  Asynchronous, ordered behind the tile's output write. */
int shp_overview_window_dev(shp_ctx *ctx, const uint32_t *d_raster, int64_t pitch, int xout, int yout,
                            int w, int h, int level, uint32_t *d_ov, int ov_w, int ov_h);
/* the band statistics the reference derives from the segment histogram (utils.estimateStatsFromHisto,
 * utils.py:47-95), evaluated as numpy evaluates them there (int64 sums, float64 pairwise sum for the
 * variance, float64 comparison for the median): out[0..5] = minimum, maximum, mean, standard deviation,
 * mode, median of hist[0..n).  Host only: no device work, no context. */
int shp_hist_stats(const uint32_t *hist, int64_t n, double *out);
/* histogram of a device label raster, hist_out_host[0..max_seg_id], entry 0 zeroed (the RAT
 * Histogram column, HistogramAccumulator tiling.py:1915-1963).  ncols = the raster's row length
 * (npix a multiple of it; lets a segment's pixels be combined per 2-D patch), or 0. */
int shp_histogram_dev(shp_ctx *ctx, const uint32_t *d_raster, int64_t npix, int64_t ncols,
                      uint32_t max_seg_id, uint32_t *hist_out_host);

/* ---- the tables of the elimination stage, exported ------------------------------------------------
 * shepseg.makeSegmentLocations (shepseg.py:880-915; RowColArray :816-870) as a CSR: segment s's pixels
 * are pix_out[offsets_out[s] .. offsets_out[s + 1]) (linear indices row * ncols + col, raster
 * order), s = 0 .. max_seg_id; offsets_out has max_seg_id + 2 entries, pix_out nrows * ncols.
 * (Entry 0 lists the null pixels; the reference's dict has no key 0.) */
int shp_segment_locations(shp_ctx *ctx, const uint32_t *seg, int nrows, int ncols, uint32_t max_seg_id,
                          uint32_t *offsets_out, uint32_t *pix_out);
/* shepseg.buildSegmentSpectra (shepseg.py:780-813): float32 per-band sums of every segment's pixels
 * accumulated in raster order, spect_sum_out[(max_seg_id + 1) * nbands], row 0 = the null pixels */
int shp_build_segment_spectra(shp_ctx *ctx, const uint32_t *seg, const void *img, int dtype, int nbands,
                              int nrows, int ncols, uint32_t max_seg_id, float *spect_sum_out);

/* ---- per-segment statistics ("tilingstats") ----------------------------------------------------
 * replaces tilingstats.accumulateSegDict / calcStatsForCompletedSegs / SegmentStats / RatPage
 * (tilingstats.py:466-617, :866-1008, :1949-2045) for one image band against a label raster.
 * stats_sel: nstats x 5 uint32 = {globalCol, statId, colType, colArrayIdx, param} exactly as
 * tilingstats.makeFastStatsSelection (:798-863) builds it; statId 0..7 = min, max, mean, stddev,
 * median, mode, percentile, pixcount; colType 0 = integer column, 1 = float column.
 * intcols_out: (#int stats) x (max_seg_id+1) int64; floatcols_out: (#float stats) x
 * (max_seg_id+1) float32 (the RatPage arrays, all pages concatenated); row 0 is zero. */
int shp_segstats(shp_ctx *ctx, const uint32_t *seg, const void *band, int dtype, int64_t npix,
                 uint32_t max_seg_id, int has_null, int64_t null_val, const uint32_t *stats_sel,
                 int nstats, int64_t missing, int64_t *intcols_out, float *floatcols_out);
/* same with the label raster and the band already in device memory */
int shp_segstats_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                     int64_t npix, uint32_t max_seg_id, int has_null, int64_t null_val,
                     const uint32_t *stats_sel, int nstats, int64_t missing,
                     int64_t *intcols_out, float *floatcols_out);
/* The same for a raster whose shape is known (nrows x ncols pixels, row-major; one tile block of
 * calcPerSegmentStatsTiled, tilingstats.py:183-206): where the average segment is at most 64 pixels the
 * statistics are computed patch by patch in LDS and only the segments that straddle patches are sorted. */
int shp_segstats2d_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                       int64_t nrows, int64_t ncols, uint32_t max_seg_id, int has_null, int64_t null_val,
                       const uint32_t *stats_sel, int nstats, int64_t missing,
                       int64_t *intcols_out, float *floatcols_out);

/* Multi-GPU split of the statistics: a segment that straddles two ranks' rows needs its
 * pixels from both.  Writes (segment id, band value) of every pixel whose segment id s has
 * flags[s] != 0 (flags: max_seg_id+1 bytes, host) to the host arrays, in no particular order;
 * *count_out = number of such pixels (when it exceeds cap only the first cap pairs are stored).
 * Plays the part of the per-segment dictionaries the reference keeps alive across tiles until
 * checkSegComplete sees the whole segment (tilingstats.py:518-553). */
int shp_gather_flagged_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                           int64_t npix, uint32_t max_seg_id, const uint8_t *flags, int64_t cap,
                           uint32_t *seg_out, int64_t *val_out, int64_t *count_out);

/* The same split with everything left in device memory (the data path of calcPerSegmentStatsDistributed
 * under RCCL; the reference has no counterpart: its per-segment dictionaries live in one process,
 * tilingstats.py:466-553).
 *  shp_dstats_local_dev: the statistics of this rank's nrows x ncols rows, then every id judged against the
 *    GLOBAL histogram d_hist (max_seg_id + 1 uint32 in device memory: the reference's segSize, :165): rows of
 *    segments complete on this rank stay, rows of straddlers (fewer pixels here than the histogram says) are
 *    cleared and their pixels packed as (id, value) pairs, rows of ids nobody holds keep the "missing" values
 *    on the one rank that passes keep_unheld != 0 -- so the ranks' columns ADD UP to the one-GPU columns.
 *    d_cols (caller's device memory): (#int stats) int64 columns then (#float stats) float32 columns of
 *    max_seg_id + 1 rows.  *d_pair_seg_out / *d_pair_val_out: the pairs (uint32 ids, int64 values) in the
 *    context's workspace, valid until its next call; *n_pairs_out of them, *n_straddlers_out segments.
 *  shp_dstats_merge_dev: after the all-gather -- `world` slots of `slot` pairs, counts[r] valid in slot r --
 *    the pairs with id_lo <= id < id_hi (*n_merged_out of them, of *n_ids_out segments) are reduced with the
 *    same code and their rows written into d_cols.
 *  Summing d_cols over the ranks (one integer all-reduce over the whole block read as int64 words: every
 *  32-bit half has at most one non-zero contributor) gives the columns shp_segstats2d_dev returns. */
int shp_dstats_local_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype, int64_t nrows,
                         int64_t ncols, uint32_t max_seg_id, int has_null, int64_t null_val,
                         const uint32_t *stats_sel, int nstats, int64_t missing, const uint32_t *d_hist,
                         int keep_unheld, void *d_cols, void **d_pair_seg_out, void **d_pair_val_out,
                         int64_t *n_pairs_out, int64_t *n_straddlers_out);
int shp_dstats_merge_dev(shp_ctx *ctx, const uint32_t *d_pair_seg, const int64_t *d_pair_val, int64_t slot, int world,
                         const uint32_t *counts, int dtype, uint32_t max_seg_id, int has_null, int64_t null_val,
                         const uint32_t *stats_sel, int nstats, int64_t missing, uint32_t id_lo, uint32_t id_hi,
                         void *d_cols, int64_t *n_merged_out, int64_t *n_ids_out);

/* ---- subset --------------------------------------------------------------------------
 * replaces the tile loop of subset.subsetImage (subset.py:124-166) and its njit kernel
 * processSubsetTile (subset.py:366-425): the window (tlx, tly, xs, ys) of a label raster is
 * recoded to ids 1..n in first-seen order, the window being visited in tiles of tile_size x
 * tile_size (the reference uses tiling.TILESIZE = 1024), tile rows outer, raster order inside a
 * tile; pixels that are null (0) or masked out (mask byte == 0; mask may be NULL) become 0.
 * out: xs*ys labels; orig_out[new id] = old id (row 0 = 0) is the reference's recodeDict
 * inverted -- the row gather that copySubsettedSegmentsToNew (:232-266) applies to every RAT
 * column and the optional origSegIdColName column (:207-226); hist_out[new id] = pixel count
 * (histogramDict).  orig_out / hist_out hold cap rows (n + 1 are written); *n_new_out = n.
 * Error "Requested subset is not within input image" as subset.py:86-88. */
int shp_subset_recode(shp_ctx *ctx, const uint32_t *seg, int64_t img_rows, int64_t img_cols,
                      int64_t tlx, int64_t tly, int64_t xs, int64_t ys, const uint8_t *mask,
                      int tile_size, uint32_t max_seg_id, uint32_t *out, uint32_t *orig_out,
                      uint32_t *hist_out, int64_t cap, uint32_t *n_new_out);
/* same with the label raster, the mask and the output window in device memory */
int shp_subset_recode_dev(shp_ctx *ctx, const uint32_t *d_seg, int64_t img_rows, int64_t img_cols,
                          int64_t tlx, int64_t tly, int64_t xs, int64_t ys, const uint8_t *d_mask,
                          int tile_size, uint32_t max_seg_id, uint32_t *d_out, uint32_t *orig_out,
                          uint32_t *hist_out, int64_t cap, uint32_t *n_new_out);

/* ---- spatial statistics --------------------------------------------------------------
 * replaces the tile loop of tilingstats.calcPerSegmentSpatialStatsTiled (tilingstats.py:1262-1390)
 * for the reference's built-in user functions, func = 0 userFuncMeanCoord (:1098-1142; params =
 * the six GDAL geotransform numbers; float columns 0, 1 = mean easting, northing), 1
 * userFuncNumEdgePixels (:1146-1216; params[0] = fourConnected; int column 0), 2 userFuncVariogram
 * (:1037-1094; params[0] = maxDist in 1..255; float columns 0..maxDist-1).  Only a segment's pixels
 * whose band value differs from null_val count (accumulateSegSpatial :1686-1699; the reference
 * insists on a nodata value, :1325-1333).  nint / nflt = number of integer / real columns of
 * colNamesAndTypes; intcols_out: nint x (max_seg_id+1) int64, floatcols_out: nflt x
 * (max_seg_id+1) float32; entries the function does not set, and segments without a valid pixel,
 * hold `missing`; row 0 is zero.  Arbitrary njit callbacks are not supported. */
int shp_spatialstats(shp_ctx *ctx, const uint32_t *seg, const void *band, int dtype, int64_t nrows,
                     int64_t ncols, uint32_t max_seg_id, int64_t null_val, int func,
                     const double *params, int64_t missing, int nint, int nflt,
                     int64_t *intcols_out, float *floatcols_out);
/* same with the label raster and the band already in device memory */
int shp_spatialstats_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                         int64_t nrows, int64_t ncols, uint32_t max_seg_id, int64_t null_val,
                         int func, const double *params, int64_t missing, int nint, int nflt,
                         int64_t *intcols_out, float *floatcols_out);

/* ---- multi-GPU exchange -----------------------------------------------------------------
 * One process per GPU.  The reference ships whole pickled tile results to one process over a
 * multiprocessing.managers TCP channel (NetworkDataChannel, tiling.py:1799-1912; SegmentationResultCache
 * :1966-2001); here the tiles are sharded and only the stitch's boundary data crosses GPUs, over RCCL:
 * shp_comm_send / shp_comm_recv move a recoded overlap strip (device memory, xGMI point to point),
 * shp_comm_bcast the k-means centres, shp_comm_allgather the sample parts, shp_comm_allreduce
 * (op 0: int64 sum, op 1: float64 max) the segment histogram and the timing.  A communicator is
 * bound to a context (its device and stream); every call returns when the operation is complete.
 * shp_comm_unique_id: 128 bytes made by rank 0 and handed to the other ranks out of band. */
typedef struct shp_comm shp_comm;
int  shp_comm_unique_id(void *id_out_128);
int  shp_comm_create(shp_ctx *ctx, int rank, int world, const void *unique_id_128, shp_comm **out);
void shp_comm_destroy(shp_comm *comm);
int  shp_comm_send(shp_comm *comm, const void *d_buf, size_t bytes, int dst);
int  shp_comm_recv(shp_comm *comm, void *d_buf, size_t bytes, int src);
int  shp_comm_bcast(shp_comm *comm, void *d_buf, size_t bytes, int root);
int  shp_comm_allgather(shp_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank);
int  shp_comm_allreduce(shp_comm *comm, void *d_buf, size_t count, int op);
/* ncclCommCount: how many ranks RCCL itself says the communicator spans (quoted by the bench line). */
int  shp_comm_count(shp_comm *comm, int *nranks_out);
/* The strips of the parallel stitch (distributed.py, replacing the whole-tile pickles of tiling.py:1799-1912)
 * travel asynchronously: shp_comm_isend enqueues the send on the communicator's own stream behind an event
 * recorded on `producer`'s stream (the chain step that writes the strip), shp_comm_irecv enqueues the
 * receive there and makes `consumer`'s stream wait for it on the device; neither waits on the host, so a
 * rank's chain never stalls for its neighbour.  Operations of one communicator complete in issue order; both
 * ends issue them in the order of the boundary plan.  shp_comm_group(1) / (0) = ncclGroupStart / End (a rank
 * that sends to itself must group the pair; inside a group pass consumer = NULL and call shp_comm_wait after
 * the group's end: the receive is only enqueued then); shp_comm_drain waits on the host for everything issued. */
int  shp_comm_isend(shp_comm *comm, const void *d_buf, size_t bytes, int dst, shp_ctx *producer);
int  shp_comm_irecv(shp_comm *comm, void *d_buf, size_t bytes, int src, shp_ctx *consumer);
int  shp_comm_group(shp_comm *comm, int begin);
int  shp_comm_wait(shp_comm *comm, shp_ctx *consumer);
int  shp_comm_drain(shp_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* SHEPSEG_HIP_H */
