/* dcahost.h -- host-side (no GPU) entry points of the MI355X DCA path: the output formats on the
 * far side of the hot path.  Plain C ABI, built with g++ into dca_amd/csrc/libdcahost.so.
 *
 * Replaces, for the result matrices the CLI writes (dca/network.py:223-231, 413-421):
 *   dca/io.py:120-129  write_text_matrix(matrix, filename, rownames, colnames, transpose)
 *     = pandas.DataFrame(matrix, index=rownames, columns=colnames).to_csv(filename, sep='\t',
 *       index=(rownames is not None), header=(colnames is not None), float_format='%.6f')
 * The bytes written are those pandas writes (tests/test_tsv_cpu.py compares them): one header line
 * (an empty first cell when there is an index column), then per row the optional name and the
 * values as correctly rounded '%.6f' (round-half-even on the exact binary value, '-0.000000'
 * keeps its sign, NaN -> empty field, +-inf -> 'inf' / '-inf'), '\n' line ends.  Names are written
 * verbatim: the caller keeps names that would need CSV quoting (tab, quote, CR, LF) away from here.
 *
 * At BASELINE configs[2] one result matrix is 20 000 x 68 579 values = 12 GB of text; the reference
 * formats it value by value in Python.  Here row blocks are formatted by a pool of threads (fp32
 * values through an exact integer path: mantissa x 15625 shifted by the exponent, no floating-point
 * rounding anywhere) and written in order by the calling thread.
 */
#ifndef DCAHOST_H
#define DCAHOST_H

#ifdef __cplusplus
extern "C" {
#endif

#define DCAHOST_OK 0
#define DCAHOST_EINVAL (-1)
#define DCAHOST_EIO (-2)
#define DCAHOST_EUNSUPPORTED (-3)   /* the reader meets something it leaves to pandas (quotes, ragged lines, text) */

/* Matrix element (r, c) of the OUTPUT is data[r * row_stride + c * col_stride] (strides in
 * elements), so transpose=True of the reference is row_stride = 1, col_stride = ld of the stored
 * matrix -- no transposed copy is made by the caller; row blocks are gathered through cache-sized
 * tiles.  rownames / colnames: arrays of nrows / ncols NUL-terminated UTF-8 strings, or NULL
 * (no index column / no header line).  nthreads <= 0: one per hardware thread (at most 64).
 * Returns DCAHOST_OK, DCAHOST_EINVAL (bad arguments) or DCAHOST_EIO (open / write failed; errno set). */
int dcahost_write_tsv_f32(const char* path, const float* data, long nrows, long ncols,
                          long row_stride, long col_stride,
                          const char* const* rownames, const char* const* colnames, int nthreads);

int dcahost_write_tsv_f64(const char* path, const double* data, long nrows, long ncols,
                          long row_stride, long col_stride,
                          const char* const* rownames, const char* const* colnames, int nthreads);

/* The same file written row block by row block: open writes the header line (colnames NULL: none; has_index: an
 * empty first cell), every rows call appends nrows rows of ncols values (row r at data + r * ld) in call order --
 * the fused predict writer (dca/network.py:407-421 written gene x cell, dca/io.py:120-129) hands over gene blocks of
 * the result as they leave the GPU, so that no cells x genes matrix is ever staged on the host.  Same bytes as
 * dcahost_write_tsv_f32 on the whole matrix. */
int dcahost_tsv_stream_open(const char* path, long ncols, const char* const* colnames, int has_index, void** handle);
int dcahost_tsv_stream_rows_f32(void* handle, const float* data, long nrows, long ld,
                                const char* const* rownames, int nthreads);
int dcahost_tsv_stream_close(void* handle);

/* Formats n values as '%.6f' separated by tabs into out (capacity cap bytes); returns the number of
 * bytes written or DCAHOST_EINVAL when cap is too small (64 bytes per value always suffice).
 * The formatting kernel of the writers, exported for the parity tests. */
long dcahost_format_f32(const float* v, long n, char* out, long cap);
long dcahost_format_f64(const double* v, long n, char* out, long cap);

/* memcpy on nthreads threads (<= 0: one per hardware thread, at most 32), page-aligned shares: moves the result
 * matrices of predict() from the pinned staging buffers into the caller's arrays (dca/network.py:188-211, 395-405). */
int dcahost_parallel_copy(void* dst, const void* src, long nbytes, int nthreads);

/* Exact 64-bit content mark of nbytes bytes (every byte takes part; independent of nthreads): decides whether device
 * tensors made from a host matrix still belong to it -- the reference always feeds the CURRENT adata.X
 * (dca/network.py:188-211), so a resident copy may only be used while the host matrix is unchanged. */
unsigned long long dcahost_checksum(const void* data, long nbytes, int nthreads);

/* The count matrix of `dca <input> <outdir>` read natively (dca/io.py:59: sc.read(filename, first_column_names=True),
 * restated as pandas.read_csv(sep, index_col=0).values.astype(float32)): one header line of column names (with or
 * without a label for the name column), then one name + ncols numbers per line; '\n' or '\r\n' line ends, blank
 * lines skipped, empty field / NA spellings = NaN.  The file is mapped and parsed by a pool of threads (nthreads <= 0:
 * one per hardware thread, at most 64, at least 4 MB of text each).
 *   dcahost_tsv_open  : maps and indexes the file; shape and the byte sizes the name buffers need.
 *   dcahost_tsv_read_f32 : values -> out[row * ld + col]; row / column names '\n'-joined, NUL-terminated.
 *   dcahost_tsv_close : unmaps.
 * DCAHOST_EUNSUPPORTED: quoted fields, ragged lines or a field that is not a number -- the caller falls back to pandas. */
int dcahost_tsv_open(const char* path, char sep, int nthreads, void** handle, long* nrows, long* ncols,
                     long* rowname_bytes, long* colname_bytes);
int dcahost_tsv_read_f32(void* handle, float* out, long ld, char* rownames, long rowname_cap,
                         char* colnames, long colname_cap);
void dcahost_tsv_close(void* handle);

#ifdef __cplusplus
}
#endif
#endif
