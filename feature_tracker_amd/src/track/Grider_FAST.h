// Grid FAST extraction (call surface of src/track/Grider_FAST.h:45-135).  The image lives in a device pyramid
// slot; one batched call detects, suppresses and ranks every cell on the GPU.
#pragma once
#include <vector>

#include "../types.h"
#include "FrontendBackend.h"

namespace feature_tracker {

// where an image lives: (camera stream, ring slot) of a FrontendBackend
struct DeviceImage {
    FrontendBackend* backend = nullptr;
    int stream = 0, slot = 0;
    int cols = 0, rows = 0;
};

class Grider_FAST {
public:
    // keypoints with the highest response first (Grider_FAST.h:58-60); equal responses are ordered by (y, x) so
    // that the result does not depend on a sort implementation
    static bool compare_response(const KeyPoint& first, const KeyPoint& second) {
        if (first.response != second.response) return first.response > second.response;
        if (first.pt.y != second.pt.y) return first.pt.y < second.pt.y;
        return first.pt.x < second.pt.x;
    }

    static int cells_for(int cols, int rows, int grid_x, int grid_y) {
        const int size_x = cols / grid_x, size_y = rows / grid_y;
        if (size_x < 1 || size_y < 1) return 0;
        return (cols / size_x) * (rows / size_y);
    }

    // Grider_FAST.h:76-132: split the image into grid_x x grid_y cells, FAST in each, keep the best
    // num_features/(grid_x*grid_y)+1 per cell, concatenate in cell order.  nonmaxSuppression must be true.
    static int perform_griding(const DeviceImage& img, std::vector<KeyPoint>& pts, int num_features, int grid_x, int grid_y,
                               int threshold, bool nonmaxSuppression) {
        if (!img.backend || !nonmaxSuppression) return FT_E_INVALID;
        const int cells = cells_for(img.cols, img.rows, grid_x, grid_y);
        if (cells < 1) return FT_E_INVALID;
        const int per_cell = num_features / (grid_x * grid_y) + 1;
        std::vector<ft_keypoint> raw((size_t)cells * per_cell);
        int n_out = 0;
        int rc = img.backend->fast_grid_async(1, &img.stream, &img.slot, threshold, grid_x, grid_y, per_cell, raw.data(),
                                              (int)raw.size(), &n_out, nullptr);
        if (rc == FT_OK) rc = img.backend->sync();
        if (rc != FT_OK) return rc;
        for (int i = 0; i < n_out; i++) pts.push_back(to_keypoint(raw[i]));
        return FT_OK;
    }

    static KeyPoint to_keypoint(const ft_keypoint& k) {
        KeyPoint kp;
        kp.pt = Point2f(k.x, k.y);
        kp.response = k.response;
        return kp;
    }
};

}  // namespace feature_tracker
