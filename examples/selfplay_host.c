/* selfplay_host.c -- the hot path driven from plain C through the C ABI (include/othellozero_amd.h), no Python, no torch:
 * one rank of a (possibly multi-GPU) self-play job.  It builds a freshly initialised OthelloNN (oz_net_init_random), plays `games` concurrent
 * 6x6 games to the end with the batched engine, and pools the move records of all ranks with the library's own RCCL communicator
 * (oz_comm_* / oz_selfplay_gather_records -- the role of WorkerManager.get_results, workers.py:168-184).
 *
 *   gcc -O2 -Iinclude examples/selfplay_host.c -o selfplay_host -Lothellozero_amd/lib -lothellozero_amd -Wl,-rpath,$PWD/othellozero_amd/lib
 *   ./selfplay_host [rank world id_file]          (one process per GPU; rank 0 writes the 128-byte communicator id to id_file,
 *                                                  the other ranks read it; without arguments: one rank)
 */
#ifndef _DEFAULT_SOURCE
#define _DEFAULT_SOURCE            /* usleep */
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "othellozero_amd.h"

#define MAX_WORLD 64
#define CHECK(call) do { int rc_ = (call); if (rc_ != OZ_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, oz_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int rank = argc > 2 ? atoi(argv[1]) : 0, world = argc > 2 ? atoi(argv[2]) : 1;
    const char* id_file = argc > 3 ? argv[3] : NULL;
    const int n = 6, channels = 128, games = 64, sims = 8;
    if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) { fprintf(stderr, "usage: %s [rank world id_file], 0 <= rank < world <= %d\n", argv[0], MAX_WORLD); return 2; }
    if (world > 1 && !id_file) { fprintf(stderr, "%d ranks need an id_file to pass the communicator id through\n", world); return 2; }
    if (oz_device_count() <= 0) { fprintf(stderr, "no HIP device\n"); return 2; }
    CHECK(oz_set_device(rank % oz_device_count()));

    /* the network: a fresh OthelloNN as Keras would initialise it, the same weights on every rank (same seed).  A trained network comes in through
     * oz_net_set_weight(net, i, data, count), i = 0 .. oz_net_num_weights(net) - 1 in Keras' get_weights() order. */
    oz_net* net = NULL;
    CHECK(oz_net_create(&net, n, channels, games));
    CHECK(oz_net_init_random(net, 2024));
    CHECK(oz_net_commit(net));

    /* games sharded by global id: rank r owns ids [r * games, (r + 1) * games); per-game RNG streams are keyed by the global id */
    oz_selfplay_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n = n; cfg.num_games = games; cfg.sims = sims; cfg.q_mode = OZ_QMODE_F64; cfg.c = 1.0; cfg.temperature = 1.0; cfg.e_greedy = 0.9;
    cfg.seed = 1234; cfg.first_game_id = (uint64_t)rank * games; cfg.game_id_stride = (uint64_t)world * games;
    oz_selfplay* sp = NULL;
    CHECK(oz_selfplay_create(&sp, &cfg, net));
    CHECK(oz_selfplay_run(sp, n * n));                        /* n*n move rounds end every game */
    CHECK(oz_selfplay_sync(sp));
    oz_selfplay_stats st;
    CHECK(oz_selfplay_get_stats(sp, &st));

    /* the exchange step: rank 0 makes the id, the host hands it over (here: a file), every rank joins */
    uint8_t id[OZ_COMM_ID_BYTES];
    memset(id, 0, sizeof id);
    if (rank == 0) {
        CHECK(oz_comm_unique_id(id));
        if (id_file) {       /* written under another name and renamed: a reader sees the whole id or no file at all */
            char tmp[4096];
            if (snprintf(tmp, sizeof tmp, "%s.tmp", id_file) >= (int)sizeof tmp) return 3;
            FILE* f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp, id_file) != 0) { fprintf(stderr, "rank 0: cannot write %s\n", id_file); return 3; }
        }
    } else {
        size_t got = 0;
        for (int tries = 0; tries < 600 && got != sizeof id; ++tries) {     /* up to a minute for rank 0 to get there */
            FILE* f = fopen(id_file, "rb");
            got = f ? fread(id, 1, sizeof id, f) : 0;
            if (f) fclose(f);
            if (got != sizeof id) usleep(100000);
        }
        if (got != sizeof id) { fprintf(stderr, "rank %d: no communicator id in %s\n", rank, id_file); return 3; }
    }
    oz_comm* comm = NULL;
    CHECK(oz_comm_create(&comm, id, rank, world));
    const int64_t room = (int64_t)world * games * 64;
    oz_record* pooled = (oz_record*)malloc(sizeof(oz_record) * (size_t)room);
    int64_t n_pooled = 0, per_rank[MAX_WORLD];
    CHECK(oz_selfplay_gather_records(sp, comm, 0, pooled, room, &n_pooled, per_rank));

    int64_t z_sum = 0, mine = 0;
    for (int64_t i = 0; i < n_pooled; ++i) { z_sum += pooled[i].z; mine += pooled[i].game_id % ((uint64_t)world * games) / games == (uint64_t)rank; }
    printf("rank %d of %d: %lld games, %lld moves, %lld expansions; pooled %lld records (this rank's: %lld = %lld), sum of z %lld\n", rank, world,
           (long long)st.games_completed, (long long)st.moves, (long long)st.expansions, (long long)n_pooled, (long long)per_rank[rank], (long long)mine,
           (long long)z_sum);
    const int ok = st.games_completed == games && per_rank[rank] == st.records && mine == st.records && n_pooled >= st.records;
    free(pooled);
    oz_comm_destroy(comm);
    oz_selfplay_destroy(sp);
    oz_net_destroy(net);
    if (!ok) { fprintf(stderr, "rank %d: record counts do not add up\n", rank); return 4; }
    printf("SELFPLAY_HOST_OK\n");
    return 0;
}
