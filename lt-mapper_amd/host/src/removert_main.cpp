// removert_main.cpp -- ROS-free stand-in for ltremovert/src/removert_main.cpp:3-12.
//   ltm_run <params_ltmapper.yaml> [--check-wrappers]
// reads the `removert:` namespace of the reference's own parameter file, runs Removerter::run() and exits
// (the reference node calls ros::spin() afterwards and never exits on its own).
#include <cstdio>
#include <exception>
#include <string>

#include "removert/Removerter.h"

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <params_ltmapper.yaml>\n", argv[0]);
        return 2;
    }
    try {
        RosParamServer::setParamFile(argv[1]);
        std::printf("\033[1;32m----> Removert Main Started (MI355X build).\033[0m\n");
        ltremovert::Removerter RMV;
        if (argc > 2 && std::string(argv[2]) == "--check-wrappers") return RMV.checkFineGrainedWrappers() ? 0 : 3;
        RMV.run();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ltm_run: %s\n", e.what());
        return 1;
    }
    return 0;
}
