/* oracle/ref_prune.cpp -- TEST INFRASTRUCTURE ONLY (compiled into oracle/_ref/libpruneref.so by oracle/Makefile).
 *
 * The REFERENCE'S OWN BundlerApp::RemoveBadPointsAndCameras (src/Bundle.cpp:4190-4261: ray-angle pruning), compiled
 * from where it lies.  src/Bundle.cpp as a whole drags in the entire application (image I/O, Ceres, option parsing ...), so
 * oracle/Makefile cuts THAT ONE FUNCTION DEFINITION out of the reference file into a scratch include under $(BUILD) (outside the
 * repository, like the scratch CLAPACK copies) -- from the line `int BundlerApp::RemoveBadPointsAndCameras(` to its closing brace --
 * and this translation unit supplies the few declarations the function touches:
 *   camera_params_t (the reference's lib/sfm-driver/sfm.h, included as is), v3_t / Vx / Vy / Vz (lib/matrix/vector.h, as is),
 *   matrix_diff / matrix_norm / matrix_scale / matrix_product (lib/matrix/matrix.h, as is; the objects are the reference's own
 *   lib/matrix/matrix.c on the vendored cblas), CLAMP / RAD2DEG (lib/imagelib/defines.h, as is), ImageKey / ImageKeyVector
 *   (src/ImageData.h: std::pair<int,int> and a vector of them), and a BundlerApp with m_ray_angle_threshold and GetKey().m_extra.
 * Every statement of the pruning loop is the reference's code.
 */
#include <vector>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <cmath>

extern "C" {
#include "sfm.h"        /* $(REF)/lib/sfm-driver: camera_params_t */
#include "vector.h"     /* $(REF)/lib/matrix:     v3_t, Vx, Vy, Vz */
#include "matrix.h"     /* $(REF)/lib/matrix:     matrix_diff, matrix_norm, matrix_scale, matrix_product */
}
#include "defines.h"    /* $(REF)/lib/imagelib:   CLAMP, RAD2DEG */

typedef std::pair<int, int> ImageKey;                /* src/ImageData.h */
typedef std::vector<ImageKey> ImageKeyVector;

struct KeypointStub { int m_extra; };

class BundlerApp {
public:
    int RemoveBadPointsAndCameras(int num_points, int num_cameras, int *added_order, camera_params_t *cameras, v3_t *points, v3_t *colors,
                                  std::vector<ImageKeyVector> &pt_views);
    KeypointStub &GetKey(int, int) { return m_key; }
    double m_ray_angle_threshold;
    KeypointStub m_key;
};

/* silence the per-point printf of the reference loop without touching its arithmetic, and observe the angles it computes: acos is
 * routed through a wrapper that returns libm's value unchanged and remembers the largest one (= the function's own max_angle) */
static int quiet_printf(const char *, ...) { return 0; }
static double g_max_angle = 0.0;
static double rec_acos(double x) { const double a = acos(x); if (a > g_max_angle) g_max_angle = a; return a; }
#define printf quiet_printf
#define acos rec_acos
#include "bundle_remove_bad_points.inc"     /* cut out of $(REF)/src/Bundle.cpp by oracle/Makefile */
#undef acos
#undef printf

/* views of point i: cameras colidx[rowptr[i] .. rowptr[i+1]); cameras[].t = camera centres.
 * prune[i] = 1 iff the reference cleared the point's view list; returns the reference's num_pruned.
 * angle_deg (optional): RAD2DEG of the reference's max_angle per point, observed by a second pass that calls the function one point at
 * a time (the function keeps max_angle to itself). */
extern "C" int ref_remove_bad_points(int num_points, int num_cameras, const int *rowptr, const int *colidx, camera_params_t *cameras,
                                     const double *points, double ray_angle_threshold, unsigned char *prune, double *angle_deg)
{
    std::vector<ImageKeyVector> pt_views(num_points);
    std::vector<int> added_order(num_cameras);
    for (int j = 0; j < num_cameras; j++) added_order[j] = j;
    for (int i = 0; i < num_points; i++)
        for (int k = rowptr[i]; k < rowptr[i + 1]; k++) pt_views[i].push_back(ImageKey(colidx[k], k - rowptr[i]));
    std::vector<v3_t> pts(num_points);
    for (int i = 0; i < num_points; i++) { Vx(pts[i]) = points[3 * i]; Vy(pts[i]) = points[3 * i + 1]; Vz(pts[i]) = points[3 * i + 2]; }
    BundlerApp app;
    app.m_ray_angle_threshold = ray_angle_threshold;
    const int n = app.RemoveBadPointsAndCameras(num_points, num_cameras, added_order.data(), cameras, pts.data(), NULL, pt_views);
    for (int i = 0; i < num_points; i++) prune[i] = (rowptr[i + 1] > rowptr[i] && pt_views[i].empty()) ? 1 : 0;
    if (angle_deg) {
        for (int i = 0; i < num_points; i++) {
            std::vector<ImageKeyVector> one(1);
            for (int k = rowptr[i]; k < rowptr[i + 1]; k++) one[0].push_back(ImageKey(colidx[k], k - rowptr[i]));
            g_max_angle = 0.0;
            (void)app.RemoveBadPointsAndCameras(1, num_cameras, added_order.data(), cameras, &pts[i], NULL, one);
            angle_deg[i] = RAD2DEG(g_max_angle);
        }
    }
    return n;
}
