// removert_main.cpp -- ROS-free stand-in for ltremovert/src/removert_main.cpp:3-12.
//   ltm_run <params_ltmapper.yaml> [--check-wrappers] [--gpus K | --logical-ranks K] [--bench K [--warmup W]]
// reads the `removert:` namespace of the reference's own parameter file, runs Removerter::run() and exits
// (the reference node calls ros::spin() afterwards and never exits on its own).
//   --gpus K           keyframes sharded over GPUs 0..K-1 of this node: one host thread + one device context per GPU, label
//                      masks and per-rank map pieces exchanged with RCCL over xGMI (removert/Comm.h)
//   --logical-ranks K  the same sharding with K ranks that all use GPU `removert/gpu_device` and exchange through host staging
//                      (LocalComm): the outputs must not depend on K -- this is how the sharding is tested on a 1-GPU box
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "removert/Removerter.h"

static int runSharded(int world, bool rccl)
{
    using namespace ltremovert;
    RosParamServer p;
    std::vector<std::shared_ptr<Comm>> comms;
    std::vector<int> devs;
    for (int r = 0; r < world; ++r) devs.push_back(rccl ? r : p.gpu_device_);
    comms = rccl ? makeRcclComms(devs) : makeLocalComms(world);
    std::vector<std::string> errors((size_t)world);
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r)
        threads.emplace_back([&, r] {
            logQuiet() = r != 0;
            try {
                auto dev = std::make_shared<Device>(p, devs[(size_t)r], comms[(size_t)r]);
                Removerter RMV(dev);
                RMV.run();
            } catch (const std::exception& e) {
                errors[(size_t)r] = e.what();
                std::fprintf(stderr, "ltm_run: rank %d: %s\n", r, e.what());
                comms[(size_t)r]->abort();
                if (rccl) {      // peers may be parked inside a collective on an aborted communicator: no orderly unwinding is possible
                    std::fprintf(stderr, "ltm_run: rank %d failed; the outputs under %s are INCOMPLETE (the other ranks' writers were cut off)\n", r,
                                 p.save_pcd_directory_.c_str());
                    std::fflush(stderr);
                    std::_Exit(1);
                }
            }
        });
    for (auto& t : threads) t.join();
    for (const auto& e : errors) if (!e.empty()) return 1;
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <params_ltmapper.yaml> [--check-wrappers] [--gpus K | --logical-ranks K]\n", argv[0]);
        return 2;
    }
    try {
        RosParamServer::setParamFile(argv[1]);
        std::printf("\033[1;32m----> Removert Main Started (MI355X build).\033[0m\n");
        std::fflush(stdout);
        int world = 1;
        bool rccl = false, check = false;
        int bench_steps = 0, bench_warmup = 1;
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "--check-wrappers") check = true;
            else if ((a == "--gpus" || a == "--logical-ranks") && i + 1 < argc) { world = std::atoi(argv[++i]); rccl = a == "--gpus"; }
            else if (a == "--bench" && i + 1 < argc) bench_steps = std::atoi(argv[++i]);
            else if (a == "--warmup" && i + 1 < argc) bench_warmup = std::atoi(argv[++i]);
            else { std::fprintf(stderr, "ltm_run: unknown argument %s\n", a.c_str()); return 2; }
        }
        if (world < 1 || world > 64) { std::fprintf(stderr, "ltm_run: rank count out of range\n"); return 2; }
        if (check) { ltremovert::Removerter RMV; return RMV.checkFineGrainedWrappers() ? 0 : 3; }
        if (bench_steps > 0) { ltremovert::Removerter RMV; return RMV.runBench(bench_steps, std::max(0, bench_warmup)); }
        if (world > 1 || rccl) return runSharded(world, rccl);
        ltremovert::Removerter RMV;
        RMV.run();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ltm_run: %s\n", e.what());
        return 1;
    }
    return 0;
}
