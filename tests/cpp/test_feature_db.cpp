// Host-logic checks of the feature database (call surface of src/feat/FeatureDatabase.h:53-327 and
// src/feat/Feature.cpp:29-89 in the reference): batched update equals one-by-one update, the query functions, the
// clean-up functions, and features handed out with remove=true outliving the database (its pooled memory).
#include <cassert>
#include <cstdio>
#include <memory>
#include <vector>

#include "feat/FeatureDatabase.h"

using namespace feature_tracker;

#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

#include <thread>

// shards: how many shards database `a` is split into (b stays the reference's single map); writers: threads that
// share every batched update of `a`
static int run(int shards, int writers) {
    // 1. batched == one by one, over many frames with ids appearing and disappearing
    FeatureDatabase a, b;
    CHECK(a.set_shards(shards) && a.shard_count() == shards);
    for (int frame = 0; frame < 300; frame++) {
        std::vector<size_t> ids;
        std::vector<float> uvn;
        for (size_t id = (size_t)frame / 3; id < (size_t)frame / 3 + 150; id++) {
            if ((id * 7 + (size_t)frame) % 11 == 0) continue;   // this feature is not seen in this frame
            ids.push_back(id + 1);
            const float u = (float)id + 0.25f * (float)frame, v = (float)frame - 0.5f * (float)id;
            uvn.insert(uvn.end(), {u, v, u * 0.01f, v * 0.01f});
        }
        const double t = 100.0 + 0.05 * frame;
        if (writers == 1) a.update_features(ids.size(), ids.data(), t, 0, uvn.data());
        else {
            std::vector<std::thread> th;
            for (int p = 0; p < writers; p++) th.emplace_back([&, p] { a.update_features(ids.size(), ids.data(), t, 0, uvn.data(), p, writers); });
            for (auto& x : th) x.join();
        }
        if (frame == 0) CHECK(!a.set_shards(2));   // not once features exist
        for (size_t i = 0; i < ids.size(); i++) b.update_feature(ids[i], t, 0, uvn[4 * i], uvn[4 * i + 1], uvn[4 * i + 2], uvn[4 * i + 3]);
    }
    CHECK(a.size() == b.size() && a.size() > 200);
    for (const auto& kv : b.get_internal_data()) {
        Feature* fa = a.get_feature(kv.first);
        CHECK(fa != nullptr && fa->featid == kv.first);
        CHECK(fa->timestamps.at(0).size() == kv.second->timestamps.at(0).size());
        for (size_t i = 0; i < fa->timestamps.at(0).size(); i++) {
            CHECK(fa->timestamps.at(0)[i] == kv.second->timestamps.at(0)[i]);
            CHECK(fa->uvs.at(0)[i] == kv.second->uvs.at(0)[i] && fa->uvs_norm.at(0)[i] == kv.second->uvs_norm.at(0)[i]);
        }
    }

    // 2. queries
    const double t_mid = 100.0 + 0.05 * 150;
    const size_t n_all = a.size();
    const std::vector<Feature*> old = a.features_not_containing_newer(t_mid);
    const std::vector<Feature*> older = a.features_containing_older(t_mid);
    const std::vector<Feature*> at = a.features_containing(t_mid);
    CHECK(!old.empty() && old.size() < n_all && !older.empty() && !at.empty());
    for (Feature* f : old) CHECK(f->timestamps.at(0).back() < t_mid);
    for (Feature* f : older) CHECK(f->timestamps.at(0).front() < t_mid);
    CHECK(a.size() == n_all);   // remove = false leaves the database alone

    // 3. removal hands the feature over: it must stay valid after the database is gone
    std::vector<Feature*> taken;
    size_t taken_obs = 0;
    {
        auto db = std::make_unique<FeatureDatabase>();
        for (int frame = 0; frame < 50; frame++) {
            const size_t ids[3] = {1, 2, 3};
            const float uvn[12] = {1.f * frame, 2, 3, 4, 5, 6.f * frame, 7, 8, 9, 10, 11.f * frame, 12};
            db->update_features(3, ids, 1.0 + frame, 1, uvn);
        }
        taken = db->features_containing_older(10.0, true);
        CHECK(taken.size() == 3 && db->size() == 0);
        Feature* again = db->get_feature(2);
        CHECK(again == nullptr);
        const size_t ids[1] = {2};
        const float uvn[4] = {0, 0, 0, 0};
        db->update_features(1, ids, 99.0, 1, uvn);          // the id cache must not resurrect the removed feature
        CHECK(db->size() == 1 && db->get_feature(2)->timestamps.at(1).size() == 1);
        Feature* one = db->get_feature(2, true);
        CHECK(one && db->size() == 0);
        taken.push_back(one);
    }
    for (Feature* f : taken) {
        for (const auto& kv : f->timestamps) taken_obs += kv.second.size();
        CHECK(f->uvs.at(1).size() == f->timestamps.at(1).size());
    }
    CHECK(taken_obs == 151);
    CHECK(taken[0]->uvs.at(1)[49][0] == 49.f || taken[1]->uvs.at(1)[49][0] == 49.f || taken[2]->uvs.at(1)[49][0] == 49.f);
    for (Feature* f : taken) delete f;

    // 4. clean-up functions and re-keying
    a.cleanup_measurements(t_mid);
    for (const auto& kv : a.get_internal_data()) CHECK(kv.second->timestamps.at(0).front() > t_mid && kv.second->num_measurements() >= 1);
    const size_t n_before = a.size();
    size_t first_id = 0;
    for (const auto& kv : a.get_internal_data()) { kv.second->to_delete = true; first_id = kv.first; break; }
    a.cleanup();
    CHECK(a.size() == n_before - 1 && a.get_feature(first_id) == nullptr);
    size_t some = 0;
    for (const auto& kv : a.get_internal_data()) { some = kv.first; break; }
    // (with several shards the new id lives in another shard than the old one: 1000001 - some is odd or even as needed)
    const size_t new_id = 1000000 + ((some % 4) == 0 ? 1 : 0);
    const size_t n_obs = a.get_feature(some)->num_measurements();
    a.rekey(some, new_id);
    CHECK(a.get_feature(some) == nullptr && a.get_feature(new_id) != nullptr && a.get_feature(new_id)->featid == new_id);
    CHECK(a.get_feature(new_id)->num_measurements() == n_obs);
    const size_t ids1[1] = {new_id};
    const float uvn1[4] = {1, 2, 3, 4};
    a.update_features(1, ids1, 1e6, 0, uvn1);                // the moved feature keeps growing in its new shard
    CHECK(a.get_feature(new_id)->num_measurements() == n_obs + 1);
    // re-keying onto an id that is taken leaves that feature in place (insert semantics, src/track/TrackBase.h:264)
    size_t other = 0;
    for (const auto& kv : a.get_internal_data()) if (kv.first != new_id) { other = kv.first; break; }
    const size_t other_obs = a.get_feature(other)->num_measurements(), n_now = a.size();
    a.rekey(new_id, other);
    CHECK(a.size() == n_now - 1 && a.get_feature(new_id) == nullptr && a.get_feature(other)->num_measurements() == other_obs);
    Feature* f = a.get_feature(other);
    f->clean_old_measurements({f->timestamps.at(0).front()});
    CHECK(f->num_measurements() == 1);
    std::printf("feature database checks passed (%zu features, %d shard(s), %d writer(s))\n", a.size(), shards, writers);
    return 0;
}

int main() {
    if (run(1, 1) || run(4, 1) || run(4, 3) || run(3, 2) || run(1, 2)) return 1;
    return 0;
}
