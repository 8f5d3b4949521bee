// kitti_info_cli.cpp -- prints what the KITTI reader (kitti.hpp) finds in a sequence directory:
//   kitti_info <sequence dir> [<ground truth file>]
// frames, cameras (K and the fourth column of P) and, with a ground-truth file, the first and last pose and the
// path length.  A smoke test of the reader on a machine that has the dataset; no images are opened.
#include <cstdio>
#include <iostream>

#include "kitti.hpp"

int main(int argc, char** argv) {
  if (argc < 2) { std::cout << "usage: kitti_info <sequence dir> [<ground truth file>]\n"; return 2; }
  try {
    VO::ParseKITTI pk;
    pk.loadFrames(argv[1]);
    pk.loadCameras(std::string(argv[1]) + "/calib.txt");
    printf("frames %zu", pk.frames_.size());
    if (!pk.frames_.empty()) printf(" first %s t=%.6f last %s t=%.6f", pk.frames_.front().left_frame_.c_str(), pk.frames_.front().time_,
                                    pk.frames_.back().right_frame_.c_str(), pk.frames_.back().time_);
    printf("\ncameras %zu\n", pk.cameras_.size());
    for (size_t i = 0; i < pk.cameras_.size(); ++i) {
      const VO::Camera& c = pk.cameras_[i];
      printf("camera %zu fx %.17g fy %.17g cx %.17g cy %.17g t %.17g %.17g %.17g\n", i, c.K_(0, 0), c.K_(1, 1), c.K_(0, 2), c.K_(1, 2),
             c.P_(0, 3), c.P_(1, 3), c.P_(2, 3));
    }
    if (argc > 2) {
      pk.loadGroundTruth(argv[2]);
      double len = 0.0;
      for (size_t i = 1; i < pk.ground_truth_poses_.size(); ++i) {
        double d2 = 0.0;
        for (int k = 4; k < 7; ++k) { const double d = pk.ground_truth_poses_[i].p[k] - pk.ground_truth_poses_[i - 1].p[k]; d2 += d * d; }
        len += std::sqrt(d2);
      }
      printf("ground_truth %zu path_length %.6f\n", pk.ground_truth_poses_.size(), len);
      if (!pk.ground_truth_poses_.empty()) {
        const double* p = pk.ground_truth_poses_.back().p;
        printf("last_pose %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
      }
    }
  } catch (const std::exception& e) {
    std::cerr << "kitti_info: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
