/* TEST INFRASTRUCTURE (oracle/_ref/removert_selfremovert): the reference's process with its multi-resolution remove / revert loop switched ON.
 *
 * The shipped Removerter::removeHighDynamicPoints() (ltremovert/src/Removerter.cpp:1580-1604) has its two selfRemovert(...) calls commented out
 * (:1582, :1586) and runs removeOnce(sess, sess, 2.5) instead, so the reference's own main can only produce the single-resolution output tree.
 * BASELINE configs[1] is the 3-resolution form.  This second main is removert_main.cpp (:1-16) with Removerter::run() (:1653-1678) written out, and in
 * it removeHighDynamicPoints() written out with exactly those two commented-out calls restored -- every function CALLED is the reference's own,
 * compiled from its unmodified sources like the rest of oracle/_ref (Makefile).  Nothing here computes anything. */
/* everything Removerter.h pulls in, first and with its access specifiers intact: the stand-in library headers and, through them and through
 * removert/utility.h, the C++ library ... */
#include "removert/RosParamServer.h"
#include "removert/Session.h"
/* ... then the class itself with its private section opened: run()'s steps and the two sessions are private members (Removerter.h:11-65) */
#define private public
#include "removert/Removerter.h"
#undef private

int main(int argc, char** argv)
{
    ros::init(argc, argv, "removert");
    ltremovert::Removerter RMV;
    /* Removerter::run(), Removerter.cpp:1653-1678 */
    RMV.loadSessionInfo();
    RMV.parseKeyframes();
    RMV.loadKeyframes();
    RMV.precleaningKeyframes(2.5);
    RMV.makeGlobalMap();
    {   /* Removerter::removeHighDynamicPoints(), :1580-1604, with :1582 and :1586 instead of :1584 and :1587 */
        ltremovert::Session& C = RMV.central_sess_;
        ltremovert::Session& Q = RMV.query_sess_;
        RMV.selfRemovert(C, RMV.repeat_removert_iter_);
        RMV.selfRemovert(Q, RMV.repeat_removert_iter_);
        C.extractHighDynPointsViaKnnDiff(C.map_global_curr_static_);
        Q.extractHighDynPointsViaKnnDiff(Q.map_global_curr_static_);
        auto map_central_high_dyn = mergeScansWithinGlobalCoordUtil(C.keyframe_scans_dynamic_, C.keyframe_poses_, C.kSE3MatExtrinsicLiDARtoPoseBase);
        auto map_query_high_dyn = mergeScansWithinGlobalCoordUtil(Q.keyframe_scans_dynamic_, Q.keyframe_poses_, Q.kSE3MatExtrinsicLiDARtoPoseBase);
        octreeDownsampling(map_central_high_dyn, map_central_high_dyn, 0.05);
        octreeDownsampling(map_query_high_dyn, map_query_high_dyn, 0.05);
        pcl::io::savePCDFileBinary(RMV.save_pcd_directory_ + "central_sess_high_dyn.pcd", *map_central_high_dyn);
        pcl::io::savePCDFileBinary(RMV.save_pcd_directory_ + "query_sess_high_dyn.pcd", *map_query_high_dyn);
    }
    RMV.parseStaticScansViaProjection();
    RMV.detectLowDynamicPoints();
    RMV.updateCurrentMap();
    RMV.parseUpdatedStaticScansViaProjection();
    RMV.parseLDScansViaProjection();
    RMV.updateScansScanwise();
    RMV.saveAllTypeOfScans();
    ros::spin();
    return 0;
}
