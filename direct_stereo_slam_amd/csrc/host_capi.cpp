// host_capi.cpp -- the pieces of the path that SURVEY.md section 8a keeps on the host by design:
// search_sc (<= 3 candidates per query, src/loop_closure/loop_detection/search_place.h:59-84).
// Plain C++; no device code.
#include "../../include/dsm_hotpath.h"

extern "C" {

// inner loop of search_sc, search_place.h:67-79: float accumulator, double products
float dsm_sc_distance(const int *a_idx, const double *a_val, int na, const int *b_idx, const double *b_val, int nb,
                      int sc_width) {
  float cur_prod = 0;
  int m = 0, n = 0;
  while (m < na && n < nb) {
    if (a_idx[m] == b_idx[n]) {
      cur_prod += a_val[m] * b_val[n];
      m++;
      n++;
    } else {
      if (a_idx[m] < b_idx[n])
        m++;
      else
        n++;
    }
  }
  const float cur_diff = (1 - cur_prod / sc_width) / 2.0;
  return cur_diff;
}

// search_sc, search_place.h:59-84: first minimal candidate wins (strict <)
int dsm_search_sc(const int *sig_idx, const double *sig_val, int n_sig, int n_cand, const int *cand_ids,
                  const int *const *cand_idx, const double *const *cand_val, const int *cand_n, int sc_width,
                  int *res_idx, float *res_diff) {
  if (n_cand < 1 || !cand_ids || !res_idx || !res_diff) return DSM_ERR_INVALID;
  *res_idx = cand_ids[0];
  *res_diff = 1.1;
  for (int c = 0; c < n_cand; c++) {
    const float cur = dsm_sc_distance(sig_idx, sig_val, n_sig, cand_idx[c], cand_val[c], cand_n[c], sc_width);
    if (*res_diff > cur) {
      *res_idx = cand_ids[c];
      *res_diff = cur;
    }
  }
  return DSM_OK;
}

} // extern "C"
