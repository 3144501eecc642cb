// Host-side sanitizer run of the image decoders over damaged copies of a JPEG file:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iray-tracing_amd/csrc/host -Iinclude \
//       tools/fuzz_jpeg.cpp ray-tracing_amd/csrc/host/jpeg.cpp -o /tmp/fuzz_jpeg && /tmp/fuzz_jpeg assets/earthmap.jpg 20000
// Every decode must end in an image or an exception; the sanitizers abort on anything else.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <sstream>
#include <stdexcept>

#include "scene.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string good = ss.str();
    const int trials = argc > 2 ? std::atoi(argv[2]) : 1000;
    std::mt19937 rng(12345);
    long decoded = 0, rejected = 0;
    for (int t = 0; t < trials; ++t) {
        std::string bad = good;
        const size_t region = (t % 3 == 0) ? bad.size() : 400;  // the headers of the usual test file end near byte 400
        const int hits = 1 + (int)(rng() % 6);
        for (int k = 0; k < hits; ++k) bad[2 + rng() % (region - 2)] = (char)(rng() & 255);
        if (t % 7 == 0) bad.resize(2 + rng() % (bad.size() - 2));
        try {
            v4ray::Image img = v4ray::decode_jpeg((const uint8_t *)bad.data(), bad.size());
            if (img.rgb.size() != (size_t)img.width * img.height * 3) return 1;
            ++decoded;
        } catch (const std::exception &) {
            ++rejected;
        }
    }
    std::printf("%d damaged files: %ld decoded, %ld rejected\n", trials, decoded, rejected);
    return 0;
}
