// Comm.hpp on the CPU: the shared-memory all-reduce between `world` processes (sums in rank order, rounds longer than the
// segment's capacity) and the rendezvous file for the RCCL id.
// usage: comm_check shm <name> <rank> <world> <count> <capacity>   -> prints "ok <checksum>"
//        comm_check file <path> <rank> [max_age_s]                 -> rank 0 publishes 128 known bytes, others print what they read
//        comm_check reopen <name> <rank> <world>                   -> open / all-reduce / close / open again on the SAME object (the barrier's sense starts over)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "admm/Comm.hpp"
using namespace admm;

int main(int argc, char **argv) {
    if (argc >= 7 && !strcmp(argv[1], "shm")) {
        const int rank = atoi(argv[3]), world = atoi(argv[4]); const long count = atol(argv[5]); const size_t cap = (size_t)atol(argv[6]);
        comm::ShmAllReduce shm;
        if (!shm.open(argv[2], rank, world, cap, 20.0)) { fprintf(stderr, "rank %d: open failed\n", rank); return 2; }
        double sum = 0.0;
        for (int round = 0; round < 3; ++round) {
            std::vector<double> v(count);
            for (long i = 0; i < count; ++i) v[i] = std::sin(0.1 * i + round) * (rank + 1) + 1e-9 * rank;
            if (!shm.allreduce(v.data(), count)) { fprintf(stderr, "rank %d: allreduce failed\n", rank); return 3; }
            for (long i = 0; i < count; ++i) {          // the expected value, added in rank order like the transport does
                double e = std::sin(0.1 * i + round) * 1 + 0.0;
                for (int r = 1; r < world; ++r) e += std::sin(0.1 * i + round) * (r + 1) + 1e-9 * r;
                if (v[i] != e) { fprintf(stderr, "rank %d: element %ld round %d: %.17g != %.17g\n", rank, i, round, v[i], e); return 4; }
                sum += v[i];
            }
        }
        printf("ok %.17g\n", sum);
        return 0;
    }
    if (argc >= 5 && !strcmp(argv[1], "reopen")) {
        const int rank = atoi(argv[3]), world = atoi(argv[4]);
        comm::ShmAllReduce shm;
        double v[3];
        for (int pass = 0; pass < 3; ++pass) {      // an odd number of barriers per pass (1 in open + 2 per all-reduce) leaves sense_ at 1: the next open must reset it
            if (!shm.open(std::string(argv[2]) + (char)('a' + pass), rank, world, 64, 20.0)) { fprintf(stderr, "rank %d: open %d failed\n", rank, pass); return 2; }
            for (int i = 0; i < 3; ++i) v[i] = rank + 1.0 + i + pass;
            if (!shm.allreduce(v, 3)) return 3;
            for (int i = 0; i < 3; ++i) { double e = 0; for (int r = 0; r < world; ++r) e += r + 1.0 + i + pass; if (v[i] != e) { fprintf(stderr, "rank %d pass %d: %g != %g\n", rank, pass, v[i], e); return 4; } }
            shm.close();
        }
        printf("ok\n");
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "file")) {
        const int rank = atoi(argv[3]); const double max_age = argc > 4 ? atof(argv[4]) : 600.0;
        unsigned char id[128];
        for (int i = 0; i < 128; ++i) id[i] = rank == 0 ? (unsigned char)(3 * i + 1) : 0;
        std::string why;
        if (!comm::rccl_id_via_file(argv[2], rank, id, 3.0, max_age, &why)) { printf("fail %s\n", why.c_str()); return 5; }
        unsigned s = 0; for (int i = 0; i < 128; ++i) s += id[i] * (unsigned)(i + 1);
        printf("ok %u\n", s);
        return 0;
    }
    return 1;
}
