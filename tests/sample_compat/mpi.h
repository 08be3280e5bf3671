/*
 * TEST-HARNESS FIXTURE — not part of the engine, not shipped in the library.
 *
 * cutensorMp/cutensorMp_contraction.cu uses MPI only to bootstrap: rank / size (:95-97), the node-local rank
 * (:83-85), one MPI_Bcast of the ncclUniqueId (:113) and MPI_Wtime (:536, :545).  This image has no MPI runtime,
 * so to build and run the UNMODIFIED sample this header supplies exactly those entry points for ranks started on
 * one node by tests/sample_compat/mpirun.sh: rank and size come from the environment (RANK / WORLD_SIZE, the
 * names torchrun uses), the broadcast goes through a file in $CTAMD_MPI_DIR.  The engine never includes this file.
 */
#ifndef SAMPLE_COMPAT_MPI_H_
#define SAMPLE_COMPAT_MPI_H_

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef int MPI_Comm;
typedef int MPI_Info;
typedef int MPI_Datatype;

#define MPI_SUCCESS           0
#define MPI_ERR_OTHER         15
#define MPI_COMM_WORLD        0
#define MPI_COMM_TYPE_SHARED  1
#define MPI_INFO_NULL         0
#define MPI_BYTE              1
#define MPI_MAX_ERROR_STRING  256

static inline int ctamd_mpi_env_int(const char* name, int fallback) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}
static inline int MPI_Init(int* argc, char*** argv) { (void)argc; (void)argv; return MPI_SUCCESS; }
static inline int MPI_Finalize(void) { return MPI_SUCCESS; }
static inline int MPI_Comm_size(MPI_Comm c, int* n) { (void)c; *n = ctamd_mpi_env_int("WORLD_SIZE", 1); return MPI_SUCCESS; }
static inline int MPI_Comm_rank(MPI_Comm c, int* r) { (void)c; *r = ctamd_mpi_env_int("RANK", 0); return MPI_SUCCESS; }
/* every rank of this fixture runs on one node, so the shared-memory communicator is the world */
static inline int MPI_Comm_split_type(MPI_Comm c, int type, int key, MPI_Info info, MPI_Comm* out) {
    (void)type; (void)key; (void)info; *out = c; return MPI_SUCCESS;
}
static inline int MPI_Comm_free(MPI_Comm* c) { (void)c; return MPI_SUCCESS; }
static inline double MPI_Wtime(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static inline int MPI_Error_string(int err, char* buf, int* len) {
    snprintf(buf, MPI_MAX_ERROR_STRING, "sample_compat MPI error %d", err);
    *len = (int)strlen(buf);
    return MPI_SUCCESS;
}
static inline int MPI_Bcast(void* data, int count, MPI_Datatype type, int root, MPI_Comm c) {
    static int seq = 0;
    int rank = 0, size = 1;
    (void)type;
    MPI_Comm_rank(c, &rank);
    MPI_Comm_size(c, &size);
    const int id = seq++;
    if (size == 1) return MPI_SUCCESS;
    const char* dir = getenv("CTAMD_MPI_DIR");
    if (dir == NULL) return MPI_ERR_OTHER;
    char path[512], tmp[544];
    snprintf(path, sizeof(path), "%s/bcast_%d", dir, id);
    if (rank == root) {
        snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        FILE* f = fopen(tmp, "wb");
        if (f == NULL) return MPI_ERR_OTHER;
        const size_t w = fwrite(data, 1, (size_t)count, f);
        fclose(f);
        if (w != (size_t)count || rename(tmp, path) != 0) return MPI_ERR_OTHER;
        return MPI_SUCCESS;
    }
    for (int tries = 0; tries < 6000; ++tries) {   /* up to 60 s */
        FILE* f = fopen(path, "rb");
        if (f != NULL) {
            const size_t r = fread(data, 1, (size_t)count, f);
            fclose(f);
            return r == (size_t)count ? MPI_SUCCESS : MPI_ERR_OTHER;
        }
        usleep(10000);
    }
    return MPI_ERR_OTHER;
}

#endif
