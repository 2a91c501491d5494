// Descriptor-based stereo tracker (call surface of src/track/TrackDescriptor.h:43-150, flow of
// src/track/TrackDescriptor.cpp:27-354) on the device library: grid FAST on both images, a 256-bit binary descriptor per
// keypoint, left -> right association and temporal association by 2-nearest-neighbour Hamming matching with ratio and
// symmetry tests.  Selected by FeatureTrackerOptions::use_klt = false, as in the reference's driver.
//
// What is this build's own definition (parity unpinned, OpenCV is not available):
//  * the descriptor: cv::ORB::compute samples a table inside OpenCV; here it is the BRIEF-style descriptor of
//    ft_compute_descriptors (same patch size, same keypoints dropped near the border);
//  * equal Hamming distances: the lower train index comes first.
// The RANSAC step of robust_match (src/track/TrackDescriptor.cpp:267-304) undistorts pts0 for BOTH point sets (:284-285):
//  * temporal matches (id0 == id1): the two sets are identical, every pair fits any skew-symmetric matrix, and
//    findFundamentalMat has nothing to reject;
//  * left -> right matches (id0 != id1): the second set is pts0 under the other camera's model.  Without distortion the
//    two sets differ by an affine map of the image plane, a three-parameter family of matrices fits every pair, and again
//    nothing is rejected; with different distortion coefficients the map is no longer affine and the reference can drop
//    matches - by how much the two lens models disagree at that pixel, not by the quality of the match.
// This class returns every match that passed the ratio and symmetry tests, provided there are at least ten (:276-277):
// the reference's outcome in the first two cases.  The third is a deviation (not reproduced: it is an artefact of the
// slip at :285, and the seeded RANSAC here voids the degenerate systems that OpenCV's SVD resolves).
// set_ransac_on_matched_pairs(true) runs the outlier rejection on the matched pairs instead (the evident intent), with
// the seeded RANSAC of ft_reject_outliers.
#pragma once
#include <vector>

#include "Grider_FAST.h"
#include "TrackBase.h"

namespace feature_tracker {

class TrackDescriptor : public TrackBase {
public:
    explicit TrackDescriptor(FrontendBackend* backend, int numfeats, int numaruco, int fast_threshold, int gridx, int gridy, double knnratio,
                             int stream_base = 0)
        : TrackBase(numfeats, numaruco), be(backend), stream0(stream_base), threshold(fast_threshold), grid_x(gridx), grid_y(gridy),
          knn_ratio(knnratio) {}

    // returns FT_OK or the first error code of the device layer (then the tracker is left as it was before the call)
    int feed_stereo(double timestamp, const Image8& img_left, const Image8& img_right, size_t cam_id_left, size_t cam_id_right) override;

    void set_ransac_on_matched_pairs(bool on) { ransac_pairs = on; }
    const std::vector<uint8_t>& get_last_descriptors(size_t cam_id) { return desc_last[cam_id]; }   // 32 bytes per active point

protected:
    typedef std::vector<uint8_t> Desc;   // rows of 32 bytes

    // src/track/TrackDescriptor.cpp:193-246
    int perform_detection_stereo(const DeviceImage& img0, const DeviceImage& img1, std::vector<KeyPoint>& pts0, std::vector<KeyPoint>& pts1,
                                 Desc& desc0, Desc& desc1, size_t cam_id0, size_t cam_id1, std::vector<size_t>& ids0, std::vector<size_t>& ids1);
    // src/track/TrackDescriptor.cpp:249-306 (ratio and symmetry tests, :309-354, run inside the backend's matcher)
    int robust_match(const std::vector<KeyPoint>& pts0, const std::vector<KeyPoint>& pts1, const Desc& desc0, const Desc& desc1, size_t id0,
                     size_t id1, std::vector<ft_match>& matches);
    // grid FAST + descriptors of one image; keypoints without a descriptor are dropped (as orb->compute drops them)
    int extract(const DeviceImage& img, std::vector<KeyPoint>& pts, Desc& desc);

    FrontendBackend* be;
    int stream0;
    int threshold, grid_x, grid_y;
    double knn_ratio;
    bool ransac_pairs = false;
    unsigned pairs_fed = 0;   // seeds the optional outlier rejection
    std::map<size_t, Desc> desc_last;
};

}  // namespace feature_tracker
