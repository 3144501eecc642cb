// v4ray-hip: command-line renderer with the reference CLI's flags (src/main.rs:18-26:
// -i/--input scene.json, -o/--output output.bmp) plus --spp, --seed, --device, --batch.
// Like the reference (src/main.rs:46-79) it renders 1-spp passes progressively, prints
// "Iter N +Ts" as they complete and rewrites the output image every 5 seconds; unlike it, it
// stops after --spp passes (the reference runs until Enter is pressed; --spp 0 does the same).
// --gpus N / --devices a,b,.. split the passes over several GPUs of the node from this one process (device k
// of N renders passes k, k+N, ...; the accumulators are summed on the host whenever an image is saved).
// --checkpoint FILE makes a long job restartable: the accumulator and pass count are written there
// with every image save, and an existing file of the same size and seed is resumed from.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "v4ray_hip.h"
#include "v4ray_host.h"

namespace {
struct Opts {
    std::string input = "scene.json", output = "output.bmp", checkpoint;
    uint32_t spp = 64, batch = 8;
    uint64_t seed = 0x5EED, perlin_seed = 1;
    int device = 0;
    std::vector<int> devices;  // --gpus / --devices; empty = {device}
    bool pretransform = false;
};

bool parse(int argc, char **argv, Opts &o) {
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", name);
                exit(2);
            }
            return argv[++i];
        };
        if (a == "-i" || a == "--input") o.input = need("--input");
        else if (a == "-o" || a == "--output") o.output = need("--output");
        else if (a == "--spp") o.spp = (uint32_t)std::strtoul(need("--spp"), nullptr, 10);
        else if (a == "--batch") o.batch = (uint32_t)std::strtoul(need("--batch"), nullptr, 10);
        else if (a == "--seed") o.seed = std::strtoull(need("--seed"), nullptr, 0);
        else if (a == "--perlin-seed") o.perlin_seed = std::strtoull(need("--perlin-seed"), nullptr, 0);
        else if (a == "--device") o.device = std::atoi(need("--device"));
        else if (a == "--gpus") {
            int n = std::atoi(need("--gpus"));
            o.devices.clear();
            for (int k = 0; k < n; ++k) o.devices.push_back(k);
        } else if (a == "--devices") {
            o.devices.clear();
            std::string list = need("--devices");
            for (size_t pos = 0; pos < list.size();) {
                size_t end = list.find(',', pos);
                if (end == std::string::npos) end = list.size();
                o.devices.push_back(std::atoi(list.substr(pos, end - pos).c_str()));
                pos = end + 1;
            }
        }
        else if (a == "--pretransform-triangles") o.pretransform = true;
        else if (a == "--checkpoint") o.checkpoint = need("--checkpoint");
        else if (a == "-h" || a == "--help") return false;
        else {
            fprintf(stderr, "unknown argument %s\n", a.c_str());
            return false;
        }
    }
    if (o.batch == 0) o.batch = 1;
    if (o.devices.empty()) o.devices.push_back(o.device);
    return true;
}

// Sum of the accumulators of all renderers (one per GPU) on the host.
int gather(const std::vector<v4r_renderer *> &rs, uint32_t w, uint32_t h, std::vector<float> &sums) {
    sums.assign((size_t)w * h * 3, 0.0f);
    std::vector<float> part(sums.size());
    for (size_t k = 0; k < rs.size(); ++k) {
        if (v4r_accum_read(rs[k], k == 0 ? sums.data() : part.data(), nullptr) != V4R_OK) {
            fprintf(stderr, "reading the accumulator failed: %s\n", v4r_last_error());
            return 1;
        }
        if (k > 0)
            for (size_t i = 0; i < sums.size(); ++i) sums[i] += part[i];
    }
    return 0;
}

int save(const std::vector<v4r_renderer *> &rs, uint32_t n, uint32_t w, uint32_t h, const std::string &path) {
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    if (rs.size() == 1) {
        if (v4r_tonemap_u8(rs[0], n, rgb.data()) != V4R_OK) {
            fprintf(stderr, "tonemap failed: %s\n", v4r_last_error());
            return 1;
        }
    } else {
        std::vector<float> sums;
        if (gather(rs, w, h, sums) != 0) return 1;
        if (v4rh_tonemap_u8(sums.data(), sums.size(), n, rgb.data()) != 0) {
            fprintf(stderr, "%s\n", v4rh_last_error());
            return 1;
        }
    }
    if (v4rh_write_image(path.c_str(), rgb.data(), w, h) != 0) {
        fprintf(stderr, "%s\n", v4rh_last_error());
        return 1;
    }
    printf("Iter %u saved\n", n);  // src/main.rs:74
    return 0;
}

int save_checkpoint(const std::vector<v4r_renderer *> &rs, uint32_t n, uint32_t w, uint32_t h, uint64_t seed, const std::string &path) {
    std::vector<float> sums;
    if (gather(rs, w, h, sums) != 0) return 1;
    if (v4rh_checkpoint_save(path.c_str(), sums.data(), w, h, n, seed) != 0) {
        fprintf(stderr, "%s\n", v4rh_last_error());
        return 1;
    }
    return 0;
}

// Resume from `path` if it holds a checkpoint of this job; returns the passes already done.
int load_checkpoint(v4r_renderer *r, uint32_t w, uint32_t h, uint64_t seed, const std::string &path, uint32_t &done) {
    FILE *probe = fopen(path.c_str(), "rb");
    if (!probe) return 0;  // nothing to resume
    fclose(probe);
    uint32_t cw = 0, ch = 0, n = 0;
    uint64_t cseed = 0;
    if (v4rh_checkpoint_load(path.c_str(), &cw, &ch, &n, &cseed, nullptr, 0) != 0) {
        fprintf(stderr, "%s\n", v4rh_last_error());
        return 1;
    }
    if (cw != w || ch != h || cseed != seed) {
        fprintf(stderr, "checkpoint %s is of another job (%ux%u seed %llu)\n", path.c_str(), cw, ch, (unsigned long long)cseed);
        return 1;
    }
    std::vector<float> sums((size_t)w * h * 3);
    if (v4rh_checkpoint_load(path.c_str(), nullptr, nullptr, nullptr, nullptr, sums.data(), sums.size()) != 0) {
        fprintf(stderr, "%s\n", v4rh_last_error());
        return 1;
    }
    if (v4r_accum_write(r, sums.data(), n) != V4R_OK) {
        fprintf(stderr, "checkpoint: %s\n", v4r_last_error());
        return 1;
    }
    done = n;
    printf("Resumed at iter %u from %s\n", n, path.c_str());
    return 0;
}
}  // namespace

int main(int argc, char **argv) {
    Opts o;
    if (!parse(argc, argv, o)) {
        fprintf(stderr,
                "usage: v4ray-hip [-i scene.json] [-o output.bmp] [--spp N (0 = until Enter)] [--batch N] [--seed S]\n"
                "                 [--perlin-seed S] [--device D | --gpus N | --devices a,b,..] [--pretransform-triangles]\n"
                "                 [--checkpoint FILE]\n");
        return 2;
    }
    v4rh_scene *scene = nullptr;
    if (v4rh_scene_load_json_file(o.input.c_str(), o.perlin_seed, &scene) != 0) {
        fprintf(stderr, "cannot load %s: %s\n", o.input.c_str(), v4rh_last_error());
        return 1;
    }
    v4rh_scene_set_options(scene, o.pretransform ? 1 : 0, 0);
    const v4r_scene_desc *desc = nullptr;
    if (v4rh_scene_flatten(scene, &desc) != 0) {
        fprintf(stderr, "cannot build scene: %s\n", v4rh_last_error());
        return 1;
    }
    std::vector<v4r_renderer *> rs;
    for (int device : o.devices) {
        v4r_renderer *one = nullptr;
        if (v4r_renderer_create(desc, device, &one) != V4R_OK) {
            fprintf(stderr, "cannot create renderer on device %d: %s\n", device, v4r_last_error());
            return 1;
        }
        rs.push_back(one);
    }
    v4r_renderer *r = rs[0];  // a resumed checkpoint goes here; the other accumulators start at zero
    const uint32_t n_dev = (uint32_t)rs.size();
    std::atomic<bool> quit{false};
    std::thread waiter;
    if (o.spp == 0) {
        printf("Press enter to exit.\n");  // src/main.rs:81
        waiter = std::thread([&] {
            std::string line;
            std::getline(std::cin, line);
            quit = true;
        });
    }
    auto start = std::chrono::steady_clock::now();
    auto last_save = start;
    uint32_t done = 0;
    int rc = 0;
    if (!o.checkpoint.empty() && load_checkpoint(r, desc->width, desc->height, o.seed, o.checkpoint, done) != 0) return 1;
    while (!quit && (o.spp == 0 || done < o.spp)) {
        uint32_t n = o.spp == 0 ? o.batch : std::min(o.batch, o.spp - done);
        // passes done .. done+n-1: device k takes every n_dev-th of them, all devices run concurrently
        bool ok = true;
        for (uint32_t k = 0; k < n_dev && ok; ++k) {
            uint32_t share = n > k ? (n - k + n_dev - 1) / n_dev : 0;
            if (share) ok = v4r_render(rs[k], o.seed, done + k, n_dev, share) == V4R_OK;
        }
        for (uint32_t k = 0; k < n_dev && ok; ++k) ok = v4r_synchronize(rs[k]) == V4R_OK;
        if (!ok) {
            fprintf(stderr, "render failed: %s\n", v4r_last_error());
            rc = 1;
            break;
        }
        done += n;
        auto now = std::chrono::steady_clock::now();
        printf("Iter %u +%llds\n", done, (long long)std::chrono::duration_cast<std::chrono::seconds>(now - start).count());  // src/main.rs:48-52
        if (now - last_save >= std::chrono::seconds(5)) {  // src/main.rs:68
            rc |= save(rs, done, desc->width, desc->height, o.output);
            if (!o.checkpoint.empty()) rc |= save_checkpoint(rs, done, desc->width, desc->height, o.seed, o.checkpoint);
            last_save = now;
        }
    }
    if (rc == 0 && done > 0) rc = save(rs, done, desc->width, desc->height, o.output);
    if (rc == 0 && done > 0 && !o.checkpoint.empty()) rc = save_checkpoint(rs, done, desc->width, desc->height, o.seed, o.checkpoint);
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    if (done > 0)
        printf("%u passes of %ux%u in %.3f s = %.2f Mpaths/s\n", done, desc->width, desc->height, secs,
               (double)desc->width * desc->height * done / secs / 1e6);
    if (waiter.joinable()) {
        if (!quit) fclose(stdin);
        waiter.detach();
    }
    for (v4r_renderer *one : rs) v4r_renderer_destroy(one);
    v4rh_scene_free(scene);
    return rc;
}
