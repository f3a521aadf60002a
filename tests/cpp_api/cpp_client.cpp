// C++ client of include/mgx.hpp (tests/test_cpp_api.py).  Host-only part: schedules, timesteps,
// MultivariateNormal.  With a scenario file as argument (written by the test: params, image, robots,
// connections, all as f64 / u8 words) it builds the world through the FactorGraph-shaped API, runs
// ticks of the 10/10 interleaved schedule with prior updates and prints every belief mean as hex
// floats, which the test compares with the oracle bit for bit.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "mgx.hpp"

static std::vector<double> read_words(std::ifstream &f, size_t n) {
    std::vector<double> v(n);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(n * sizeof(double)));
    if (!f) throw std::runtime_error("scenario file truncated");
    return v;
}

int main(int argc, char **argv) {
    using namespace mgx;
    // ---- host-only ---------------------------------------------------------------------------------
    auto sched = schedule(GbpSchedule::InterleaveEvenly, 10, 10);
    if (sched.size() != 10) return 1;
    for (auto s : sched)
        if (!s.internal || !s.external) return 2;
    if (get_variable_timesteps(45, 3).size() != 16) return 3;
    auto n1 = MultivariateNormal::from_information_and_precision({1, 2, 3}, {1, 0, 0, 0, 1, 0, 0, 0, 1}, 3, 3);
    auto n2 = MultivariateNormal::from_information_and_precision({3, 2, 1}, {1, 0, 0, 0, 1, 0, 0, 0, 1}, 3, 3);
    auto sum = n1 + n2;  // lib.rs:593-617
    if (sum.mean() != std::vector<double>({8, 8, 8})) return 4;
    try {
        MultivariateNormal::from_mean_and_covariance({1, 2, 3}, {1, 0, 0, 0, 0, 0, 0, 0, 1}, 3, 3);
        return 5;
    } catch (const Error &e) {
        if (e.code != MGX_MVN_ERR_SINGULAR_COVARIANCE) return 6;
    }
    std::printf("host ok\n");
    if (argc < 2) return 0;

    // ---- device: the scenario of the test ----------------------------------------------------------
    std::ifstream f(argv[1], std::ios::binary);
    auto hdr = read_words(f, 12);  // sigma x4, safety multiplier, enable mask, n robots, K, n connections, image w, h, world size
    mgx_params p{};
    p.sigma_dynamics = hdr[0]; p.sigma_interrobot = hdr[1]; p.sigma_obstacle = hdr[2]; p.sigma_tracking = hdr[3];
    p.safety_multiplier = hdr[4]; p.tracking_switch_padding = 1.0; p.tracking_attraction_distance = 2.0;
    p.enable_mask = (uint32_t)hdr[5];
    const size_t n = (size_t)hdr[6], K = (size_t)hdr[7], n_conn = (size_t)hdr[8], iw = (size_t)hdr[9], ih = (size_t)hdr[10];
    try {
        World world(p);
        std::vector<uint8_t> rgb(iw * ih * 3);
        f.read(reinterpret_cast<char *>(rgb.data()), (std::streamsize)rgb.size());
        world.set_sdf(rgb, (uint32_t)iw, (uint32_t)ih, hdr[11], hdr[11]);
        std::vector<FactorGraph> graphs;
        std::vector<std::array<double, 2>> goals;
        std::vector<double> time_scale;
        for (size_t r = 0; r < n; r++) {
            auto m = read_words(f, 4 * K), pd = read_words(f, K), dt = read_words(f, K - 1), rest = read_words(f, 4);  // radius, goal x, goal y, time scale
            std::vector<Vector4> mean0(K);
            for (size_t i = 0; i < K; i++) std::memcpy(mean0[i].data(), &m[4 * i], sizeof(Vector4));
            graphs.push_back(world.add_robot(mean0, pd, dt, rest[0], (uint64_t)r));
            goals.push_back({rest[1], rest[2]});
            time_scale.push_back(rest[3]);
        }
        RobotNumberGenerator numbers;
        auto conns = read_words(f, 2 * n_conn);
        for (size_t c = 0; c < n_conn; c++) world.connect(graphs[(size_t)conns[2 * c]], graphs[(size_t)conns[2 * c + 1]], numbers);
        auto tail = read_words(f, 3);  // max speed, delta_t, ticks
        std::vector<int32_t> ids(n);
        for (size_t r = 0; r < n; r++) ids[r] = graphs[r].id();
        for (int tick = 0; tick < (int)tail[2]; tick++) {
            world.update_priors(ids, goals, time_scale, tail[0], tail[1]);
            world.iterate_gbp_v2(sched);
        }
        graphs[0].set_antenna_active(false);
        world.iterate_gbp_v2(sched);
        for (size_t r = 0; r < n; r++)
            for (size_t i = 0; i < K; i++) {
                const Belief b = *graphs[r].get_variable((uint32_t)i);
                std::printf("%a %a %a %a\n", b.mean[0], b.mean[1], b.mean[2], b.mean[3]);
            }
        const MessageCount mc = graphs[1].message_count();
        std::printf("counts %llu %llu %llu %llu\n", (unsigned long long)mc.sent_internal, (unsigned long long)mc.sent_external,
                    (unsigned long long)mc.received_internal, (unsigned long long)mc.received_external);
        if (graphs[0].get_variable((uint32_t)K)) return 7;  // out of range: None
    } catch (const Error &e) {
        if (e.code == MGX_ERR_NO_DEVICE) { std::printf("no gpu\n"); return 0; }
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 8;
    }
    return 0;
}
