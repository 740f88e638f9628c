// System::shard.from_env(): rank / world / local rank / rendezvous file from what a launcher exports (no GPU touched).
#include <cstdio>
#include "admm/System.hpp"
int main() {
    admm::System s;
    const int local = s.shard.from_env();
    std::printf("local %d rank %d world %d file %s\n", local, s.shard.rank, s.shard.world, s.shard.rccl_id_file.c_str());
    return 0;
}
