// fir_internal.h — what the FIR translation units call in one another (C++ linkage, library-private).
#pragma once
#include <hip/hip_runtime.h>
#include "art_internal.h"

int  artfir_general (const ArtFirArgs &a, const ArtSegTable &segs, hipStream_t st);                 // fir_general.hip; -1: span does not fit the LDS
void artfir_strict (const ArtFirArgs &a, const ArtSegTable &segs, int precise, hipStream_t st);      // fir_general.hip
bool artfir_takes_matrix_path (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref);       // fir_matrix.hip | fir_matrix64.hip
int  artfir_matrix (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref, void *stream);    // fir_matrix.hip | fir_matrix64.hip
size_t artfir_split_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref);           // fir_matrix.hip | fir_matrix64.hip (0)
bool artfir_matrix_spans_segments (const ArtFirArgs *a, const ArtSegTable *segs, int kernel_pref);  // fir_matrix.hip | fir_matrix64.hip (never)
size_t artfir_planes_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref);          // fir_matrix.hip | fir_matrix64.hip (0: no fixed-point path)
size_t artfir_rows_bytes (const ArtFirArgs *a, unsigned int outputs, int kernel_pref);            // fir_matrix.hip | fir_matrix64.hip (0: no rows kept)
void artfir_rows_touch (const ArtFirArgs *a, const ArtSegTable *segs);                               // fir_matrix.hip | fir_matrix64.hip (nothing)
