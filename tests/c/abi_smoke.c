/* The C ABI bound from plain C (no Python, no C++): include/pifuhd_hip.h must compile as C99, the library must link, report
 * the header's ABI version, refuse bad arguments with an error code + message instead of crashing, and its host-only entry
 * point (the OBJ writer, mesh_util.py:189-198) must work without a GPU.  Built and run by tests/test_abi.py. */
#include <stdio.h>
#include <string.h>

#include "pifuhd_hip.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    if (pifu_abi_version() != PIFU_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", pifu_abi_version(), PIFU_ABI_VERSION); return 1; }
    /* a null context is an error code and a message, never a crash */
    if (pifu_reserve(NULL, 1) != PIFU_ERR_INVALID || strlen(pifu_last_error()) == 0) return 1;
    if (pifu_allgather_f32(NULL, NULL, NULL, 0, NULL) != PIFU_ERR_INVALID) return 1;
    if (pifu_set_host_threads(-1) != PIFU_ERR_INVALID || pifu_set_host_threads(2) != PIFU_OK) return 1;
    const double verts[9] = {0.0, 0.5, -1.25, 1.0, 2.0, 3.0, -0.00004, 0.00005, 10.0};
    const double cols[9] = {0.1, 0.2, 0.3, 0.5, 0.5, 0.5, 1.0, 0.0, 0.25};
    const int32_t faces[3] = {0, 1, 2};
    if (pifu_write_obj(argv[1], verts, cols, 3, faces, 1) != PIFU_OK) { fprintf(stderr, "%s\n", pifu_last_error()); return 1; }
    /* the two-step writer: faces first, vertices when the colours exist */
    void *w = NULL;
    char path2[4096];
    snprintf(path2, sizeof path2, "%s.2", argv[1]);
    if (pifu_obj_begin(path2, faces, 1, 3, 1, &w) != PIFU_OK || pifu_obj_finish(w, verts, cols, 3) != PIFU_OK) return 1;
    printf("abi %d ok\n", pifu_abi_version());
    return 0;
}
