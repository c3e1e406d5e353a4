/* oracle/ref_fmatrix_stubs.c -- TEST INFRASTRUCTURE ONLY: link-time stand-ins for the image / polynomial helpers that
 * lib/imagelib/fmatrix.c references from functions the F-matrix harness (oracle/ref_fmatrix.c) never calls. */
#include <stdio.h>
#include <stdlib.h>

#define STUB(name) void name(void) { fprintf(stderr, "[ref_fmatrix] unexpected call of " #name "\n"); abort(); }
STUB(img_free) STUB(img_pixel_is_valid) STUB(img_resample_bbox) STUB(new_transform_vector) STUB(transform_free)
STUB(poly_deriv) STUB(poly_diff) STUB(poly_find_root) STUB(poly_free) STUB(poly_new) STUB(poly_product)
STUB(poly_set_coeff) STUB(poly_sum)
void f_exit(void) {}
