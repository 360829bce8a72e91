// icp_test_runner: the reference's experiment driver surface (DCReg/src/icp_main.cpp, loadConfig /
// TestRunner::runAllTests / runMethod / runSingleTest / statistics / report writers of
// DCReg/src/icp_test_runner.cpp:20-516, 604-1030, 1386-1500) on top of the C-ABI of include/dcreg.h.
// YAML keys, method-name dispatch, std::map method order, enum strings and the on-disk formats are kept, so
// DCReg/config/icp.yaml, icp_iter.yaml and icp_pk01.yaml run unmodified apart from paths.
// Additive keys (all optional): icp.use_weight_derivative, icp.always_compute_schur, icp.use_so3_parameterization (the
// reference's Config field, utils.hpp:170, which its loader never reads; false selects the Euler / LOAM engine),
// icp.euler_exact_jacobian (default false = that engine's row as the reference writes it, :2299-2346; true = the exact derivative), device, test.seed,
// test.perturb_trans_m, test.perturb_rot_deg (seeded per-run perturbation of initial_noise; the reference has no RNG), icp.fast_plane_fit
// (default false HERE: the driver reproduces the reference's reports, so its plane fit is the Eigen-shaped factorisation step for step;
// the library's own default is the reduced-instruction fit, which agrees to a few ulp - DESIGN.md).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/dcreg.h"
#include "pcd_io.hpp"
#include "yaml_lite.hpp"

namespace fs = std::filesystem;

namespace {

struct Pose6D { double roll = 0, pitch = 0, yaw = 0, x = 0, y = 0, z = 0; };   // utils.hpp:50-64
inline double deg2rad(double d) { return d * M_PI / 180.0; }
inline double rad2deg(double r) { return r * 180.0 / M_PI; }

struct RunnerConfig {                 // ICPRunner::Config, utils.hpp:132-171
    int num_runs = 1;
    bool save_pcd = true, save_error_pcd = true, visualize = false;
    std::string folder_path, source_pcd, target_pcd, output_folder;
    double error_threshold = 0.05;
    int normal_nn = 5;
    Pose6D initial_noise, gt_pose;
    double initial_matrix[16], gt_matrix[16];
    dcreg_config core;
    std::map<std::string, std::pair<std::string, std::string>> test_methods;
    int device = 0;
    uint64_t seed = 0;
    double perturb_trans = 0.0, perturb_rot_deg = 0.0;
    bool use_so3_parameterization = true;   // utils.hpp:170 (false -> the Euler / LOAM engine, :449-458)
    bool fast_plane_fit = false;            // reference-parity mode by default (see the header comment)
};

struct TestResult {                   // utils.hpp:253-303
    std::string method_name;
    bool converged = false;
    int iterations = 0;
    double time_ms = 0, trans_error_m = 0, rot_error_deg = 0, final_rmse = 0, final_fitness = 0;
    double p2p_rmse = 0, p2p_fitness = 0, chamfer_distance = 0;
    int64_t corr_num = 0;
    double final_transform[16];
    std::vector<dcreg_iter_log> iteration_data;
};

struct MethodStatistics {             // utils.hpp:306-336
    int total_runs = 0, converged_runs = 0;
    int64_t corr_num = 0;
    double mean_trans_error = 0, mean_rot_error = 0, mean_time_ms = 0, mean_iterations = 0, mean_rmse = 0, mean_fitness = 0;
    double mean_p2p_rmse = 0, mean_p2p_fitness = 0, mean_chamfer = 0;
    double std_trans_error = 0, std_rot_error = 0, std_time_ms = 0;
    double min_trans_error = std::numeric_limits<double>::max(), max_trans_error = 0;
    double min_rot_error = std::numeric_limits<double>::max(), max_rot_error = 0;
    double success_rate = 0;
};

const std::map<std::string, int> kDetection = {       // stringToDetectionMethod, :179-194
    {"NONE_DETE", DCREG_NONE_DETE}, {"SCHUR_CONDITION_NUMBER", DCREG_SCHUR_CONDITION_NUMBER},
    {"FULL_EVD_MIN_EIGENVALUE", DCREG_FULL_EVD_MIN_EIGENVALUE}, {"EVD_SUB_CONDITION", DCREG_EVD_SUB_CONDITION},
    {"FULL_SVD_CONDITION", DCREG_FULL_SVD_CONDITION}};
const std::map<std::string, int> kHandling = {        // stringToHandlingMethod, :196-215
    {"NONE_HAND", DCREG_NONE_HAND}, {"STANDARD_REGULARIZATION", DCREG_STANDARD_REGULARIZATION},
    {"ADAPTIVE_REGULARIZATION", DCREG_ADAPTIVE_REGULARIZATION}, {"PRECONDITIONED_CG", DCREG_PRECONDITIONED_CG},
    {"SOLUTION_REMAPPING", DCREG_SOLUTION_REMAPPING}, {"TRUNCATED_SVD", DCREG_TRUNCATED_SVD}};

bool in_scope(const std::string &m) {                 // dispatch on the method NAME, :438-440
    return m == "Ours" || m == "NONE" || m == "ME-SR" || m == "FCN-SR" || m == "ME-TSVD" || m == "ME-TReg";
}

Pose6D read_pose(const yamlite::Node &n) {            // :53-72 (degrees -> radians at load)
    Pose6D p;
    p.x = n["x"].as_double(); p.y = n["y"].as_double(); p.z = n["z"].as_double();
    p.roll = deg2rad(n["roll_deg"].as_double()); p.pitch = deg2rad(n["pitch_deg"].as_double()); p.yaw = deg2rad(n["yaw_deg"].as_double());
    return p;
}

bool loadConfig(const std::string &filename, RunnerConfig &c) {     // :20-153
    yamlite::Node y;
    try { y = yamlite::load_file(filename); } catch (const std::exception &e) { std::cerr << "Error loading YAML config: " << e.what() << std::endl; return false; }
    dcreg_default_config(&c.core);
    try {
        if (y.has("test")) {
            const auto &t = y["test"];
            c.num_runs = t["num_runs"].as_int(); c.save_pcd = t["save_pcd"].as_bool();
            c.save_error_pcd = t["save_error_pcd"].as_bool(); c.visualize = t["visualize"].as_bool();
            if (t.has("seed")) c.seed = (uint64_t)t["seed"].as_double();
            if (t.has("perturb_trans_m")) c.perturb_trans = t["perturb_trans_m"].as_double();
            if (t.has("perturb_rot_deg")) c.perturb_rot_deg = t["perturb_rot_deg"].as_double();
        }
        if (y.has("paths")) {
            const auto &p = y["paths"];
            c.folder_path = p["folder_path"].as_string(); c.source_pcd = p["source_pcd"].as_string();
            c.target_pcd = p["target_pcd"].as_string(); c.output_folder = p["output_folder"].as_string();
        }
        if (y.has("icp")) {
            const auto &i = y["icp"];
            if (i.has("fast_plane_fit")) c.fast_plane_fit = i["fast_plane_fit"].as_bool();
            c.core.search_radius = i["search_radius"].as_double(); c.core.max_iterations = i["max_iterations"].as_int();
            c.normal_nn = i["normal_nn"].as_int(); c.error_threshold = i["error_threshold"].as_double();
            c.core.CONVERGENCE_THRESH_TRANS = i["CONVERGENCE_THRESH_TRANS"].as_double();
            c.core.CONVERGENCE_THRESH_ROT = i["CONVERGENCE_THRESH_ROT"].as_double();
            if (i.has("use_weight_derivative")) c.core.use_weight_derivative = i["use_weight_derivative"].as_bool();
            if (i.has("always_compute_schur")) c.core.always_compute_schur = i["always_compute_schur"].as_bool();
            if (i.has("euler_exact_jacobian")) c.core.euler_exact_jacobian = i["euler_exact_jacobian"].as_bool();
            if (i.has("use_so3_parameterization")) c.use_so3_parameterization = i["use_so3_parameterization"].as_bool();
            std::cout << "CONVERGENCE_THRESH_TRANS: " << c.core.CONVERGENCE_THRESH_TRANS << std::endl;
            std::cout << "CONVERGENCE_THRESH_ROT: " << c.core.CONVERGENCE_THRESH_ROT << std::endl;
        }
        if (y.has("device")) c.device = y["device"].as_int();
        if (y.has("initial_noise")) c.initial_noise = read_pose(y["initial_noise"]);
        if (y.has("gt_pose")) c.gt_pose = read_pose(y["gt_pose"]);
        if (y.has("degeneracy")) {
            c.core.DEGENERACY_THRES_COND = y["degeneracy"]["condition_threshold"].as_double();
            c.core.DEGENERACY_THRES_EIG = y["degeneracy"]["eigenvalue_threshold"].as_double();
        }
        if (y.has("method_params")) {
            const auto &m = y["method_params"];
            if (m.has("adaptive_reg")) c.core.ADAPTIVE_REG_ALPHA = m["adaptive_reg"]["alpha"].as_double();
            if (m.has("standard_reg")) c.core.STD_REG_GAMMA = m["standard_reg"]["gamma"].as_double();
            if (m.has("pcg")) {
                c.core.KAPPA_TARGET = m["pcg"]["kappa_target"].as_double();
                c.core.PCG_TOLERANCE = m["pcg"]["tolerance"].as_double();
                c.core.PCG_MAX_ITER = m["pcg"]["max_iter"].as_int();
            }
        }
        if (y.has("test_methods"))
            for (const auto &kv : y["test_methods"].kids) {
                if (!kv.second->is_list || kv.second->list.size() < 2) throw std::runtime_error("test_methods." + kv.first + " must be [detection, handling]");
                c.test_methods[kv.first] = {kv.second->list[0], kv.second->list[1]};
            }
    } catch (const std::exception &e) { std::cerr << "Error loading YAML config: " << e.what() << std::endl; return false; }
    dcreg_pose6d_to_matrix(c.initial_noise.roll, c.initial_noise.pitch, c.initial_noise.yaw, c.initial_noise.x, c.initial_noise.y, c.initial_noise.z, c.initial_matrix);
    dcreg_pose6d_to_matrix(c.gt_pose.roll, c.gt_pose.pitch, c.gt_pose.yaw, c.gt_pose.x, c.gt_pose.y, c.gt_pose.z, c.gt_matrix);
    for (int i = 0; i < 16; ++i) c.core.gt_matrix[i] = c.gt_matrix[i];
    std::cout << "\n=== Loaded Configuration ===" << std::endl;
    std::cout << "STD_REG_GAMMA: " << c.core.STD_REG_GAMMA << std::endl;
    std::cout << "ADAPTIVE_REG_ALPHA: " << c.core.ADAPTIVE_REG_ALPHA << std::endl;
    std::cout << "KAPPA_TARGET: " << c.core.KAPPA_TARGET << std::endl;
    std::cout << "DEGENERACY_THRES_COND: " << c.core.DEGENERACY_THRES_COND << std::endl;
    std::cout << "DEGENERACY_THRES_EIG: " << c.core.DEGENERACY_THRES_EIG << std::endl;
    std::cout << "USE_WEIGHT_DERIVATIVE: " << c.core.use_weight_derivative << std::endl;
    std::cout << "USE_SO3 ICP: " << c.use_so3_parameterization << std::endl;                   // :145
    std::cout << "==========================\n" << std::endl;
    return true;
}

class TestRunner {
public:
    explicit TestRunner(const RunnerConfig &c) : config_(c) {}
    ~TestRunner() { dcreg_backend_destroy(ctx_); }

    bool runAllTests() {                                            // :299-328
        if (!loadPointClouds()) return false;
        if (dcreg_backend_create(&ctx_, config_.device) != DCREG_OK) {
            std::cerr << "[dcreg] no usable MI355X device " << config_.device << " (there is no CPU fallback)" << std::endl;
            return false;
        }
        dcreg_set_option(ctx_, "fast_plane_fit", config_.fast_plane_fit ? 1.0 : 0.0);
        for (const auto &kv : config_.test_methods) {
            const std::string &name = kv.first;
            std::cout << "\n--- Testing method: " << name << " ---" << std::endl;
            if (!in_scope(name)) {
                std::cout << "Method '" << name << "' belongs to an engine outside this build's scope (X-ICP / SuperLoc / Open3D); skipped." << std::endl;
                continue;
            }
            auto d = kDetection.find(kv.second.first);
            auto h = kHandling.find(kv.second.second);
            if (d == kDetection.end() || h == kHandling.end()) { std::cerr << "Unknown detection/handling method for " << name << std::endl; return false; }
            if (!runMethod(name, d->second, h->second)) { std::cerr << "Failed to run method: " << name << std::endl; return false; }
        }
        finalizeStatistics();
        saveStatistics();
        saveDetailedResults();
        return true;
    }

private:
    RunnerConfig config_;
    dcreg_ctx *ctx_ = nullptr;
    pcdio::Cloud source_, target_;
    std::map<std::string, MethodStatistics> statistics_;
    std::map<std::string, std::vector<TestResult>> detailed_results_;

    bool loadPointClouds() {                                        // :156-176
        std::string err;
        if (!pcdio::load(config_.folder_path + config_.source_pcd, source_, &err)) { std::cerr << "Failed to load source cloud: " << err << std::endl; return false; }
        if (!pcdio::load(config_.folder_path + config_.target_pcd, target_, &err)) { std::cerr << "Failed to load target cloud: " << err << std::endl; return false; }
        if (source_.empty() || target_.empty()) { std::cerr << "Error: Loaded point cloud is empty" << std::endl; return false; }
        std::cout << "Loaded point clouds - Source: " << source_.size() << " points, Target: " << target_.size() << " points" << std::endl;
        return true;
    }

    bool runMethod(const std::string &name, int det, int hand) {   // :331-390
        statistics_[name] = MethodStatistics();
        for (int run = 0; run < config_.num_runs; ++run) {
            if (config_.num_runs > 1 && run % 10 == 0) std::cout << "  Run " << run + 1 << "/" << config_.num_runs << std::endl;
            TestResult r = runSingleTest(name, det, hand, run);
            detailed_results_[name].push_back(r);
            updateStatistics(name, r);
            if (run == 0 && (config_.save_pcd || config_.save_error_pcd)) {          // :352-380
                std::vector<float> aligned(source_.xyzi);
                transformCloud(r.final_transform, aligned);
                if (config_.save_error_pcd) saveErrorPointCloud(aligned, config_.output_folder + name + "_error.pcd");
                if (!config_.save_pcd) continue;
                saveAlignedClouds(aligned, config_.output_folder + name + "_aligned_clouds.pcd");
                std::cout << "Saved aligned clouds for " << name << " to " << config_.output_folder + name + "_aligned_clouds.pcd" << std::endl;
                pcdio::save_binary(config_.output_folder + name + "_aligned_clouds_sig.pcd", aligned.data(), source_.size());
                std::vector<float> initial(source_.xyzi);
                transformCloud(config_.initial_matrix, initial);
                pcdio::save_binary(config_.output_folder + "initial_clouds.pcd", initial.data(), source_.size());
                pcdio::save_binary(config_.output_folder + "target_clouds.pcd", source_.xyzi.data(), source_.size());   // sic, :372-373
            }
        }
        return true;
    }

    // saveAlignedClouds, :519-552: aligned source in orange (245,121,0) followed by the target in blue-grey (144,159,207)
    void saveAlignedClouds(const std::vector<float> &aligned_xyzi, const std::string &filename) {
        std::vector<float> xyz;
        std::vector<uint32_t> rgb;
        xyz.reserve(3 * (source_.size() + target_.size()));
        for (size_t i = 0; i < source_.size(); ++i) {
            xyz.insert(xyz.end(), {aligned_xyzi[4 * i], aligned_xyzi[4 * i + 1], aligned_xyzi[4 * i + 2]});
            rgb.push_back(pcdio::pack_rgb(245, 121, 0));
        }
        for (size_t i = 0; i < target_.size(); ++i) {
            xyz.insert(xyz.end(), {target_.xyzi[4 * i], target_.xyzi[4 * i + 1], target_.xyzi[4 * i + 2]});
            rgb.push_back(pcdio::pack_rgb(144, 159, 207));
        }
        pcdio::save_binary_rgb(filename, xyz, rgb);
    }

    // saveErrorPointCloud / createErrorPointCloud / getJetColorForError, :555-600 + utils.hpp:591-627: nearest-target
    // distance of every aligned point (1-NN sweep on the device index), jet colour map clipped at
    // min(error_threshold, largest distance)
    void saveErrorPointCloud(const std::vector<float> &aligned_xyzi, const std::string &filename) {
        const size_t n = source_.size();
        std::vector<int32_t> idx(n);
        std::vector<float> d2(n);
        if (dcreg_knn(ctx_, aligned_xyzi.data(), (int64_t)n, 4, 1, 0.0, idx.data(), d2.data()) != DCREG_OK) {
            std::cerr << "error cloud: " << dcreg_last_error(ctx_) << std::endl;
            return;
        }
        double mx = 0.0;
        for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::sqrt((double)d2[i]));
        const double cmax = std::min(config_.error_threshold, mx);
        std::vector<float> xyz(3 * n);
        std::vector<uint32_t> rgb(n);
        for (size_t i = 0; i < n; ++i) {
            const double e = std::min(std::sqrt((double)d2[i]) / cmax, 1.0);
            double r, g, b;
            if (e < 0.25) { r = 0.0; g = e / 0.25; b = 1.0; }
            else if (e < 0.5) { r = 0.0; g = 1.0; b = 1.0 - (e - 0.25) / 0.25; }
            else if (e < 0.75) { r = (e - 0.5) / 0.25; g = 1.0; b = 0.0; }
            else { r = 1.0; g = 1.0 - (e - 0.75) / 0.25; b = 0.0; }
            rgb[i] = pcdio::pack_rgb((uint8_t)(255 * r), (uint8_t)(255 * g), (uint8_t)(255 * b));
            for (int k = 0; k < 3; ++k) xyz[3 * i + k] = aligned_xyzi[4 * i + k];
        }
        pcdio::save_binary_rgb(filename, xyz, rgb);
        std::cout << "Saved error visualization to " << filename << std::endl;
    }

    static void transformCloud(const double T[16], std::vector<float> &xyzi) {   // pcl::transformPointCloud<PointT,double>
        for (size_t i = 0; i < xyzi.size() / 4; ++i) {
            const double x = xyzi[4 * i], y = xyzi[4 * i + 1], z = xyzi[4 * i + 2];
            xyzi[4 * i] = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
            xyzi[4 * i + 1] = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
            xyzi[4 * i + 2] = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
        }
    }

    TestResult runSingleTest(const std::string &name, int det, int hand, int run) {   // :393-516
        TestResult r;
        r.method_name = name;
        double T0[16];
        for (int i = 0; i < 16; ++i) T0[i] = config_.initial_matrix[i];
        Pose6D p = config_.initial_noise;
        if (run > 0 && (config_.perturb_trans > 0.0 || config_.perturb_rot_deg > 0.0)) {   // additive: seeded perturbation
            // the shared generator of the C-ABI (run 0 = the unperturbed reference run); same trial -> same pose in every driver
            const double base[6] = {p.x, p.y, p.z, p.roll, p.pitch, p.yaw};
            double q[6];
            dcreg_trial_pose(base, (uint64_t)config_.seed, (int64_t)run, config_.perturb_trans, deg2rad(config_.perturb_rot_deg), T0, q);
            p.x = q[0]; p.y = q[1]; p.z = q[2]; p.roll = q[3]; p.pitch = q[4]; p.yaw = q[5];   // the Euler engine starts from the Pose6D
        }
        // context per run: index build + source upload, outside the timed region like :408-409
        if (dcreg_set_target(ctx_, target_.xyzi.data(), (int64_t)target_.size(), 4, config_.core.search_radius) != DCREG_OK ||
            dcreg_set_source(ctx_, source_.xyzi.data(), (int64_t)source_.size(), 4) != DCREG_OK) {
            std::cerr << "[ICP Error] " << dcreg_last_error(ctx_) << std::endl;
            return r;
        }
        double R0[9], t0[3];
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R0[i * 3 + j] = T0[i * 4 + j]; t0[i] = T0[i * 4 + 3]; }
        std::vector<dcreg_iter_log> log((size_t)std::max(config_.core.max_iterations, 1));
        dcreg_icp_result res;
        const auto start = std::chrono::high_resolution_clock::now();                      // :442
        const double pose6d[6] = {p.roll, p.pitch, p.yaw, p.x, p.y, p.z};
        const int rc = config_.use_so3_parameterization                                      // :443-458
                           ? dcreg_icp_run(ctx_, R0, t0, det, hand, &config_.core, log.data(), (int)log.size(), &res)
                           : dcreg_icp_run_euler(ctx_, pose6d, det, hand, &config_.core, log.data(), (int)log.size(), &res, nullptr);
        const auto end = std::chrono::high_resolution_clock::now();                        // :459
        if (rc != DCREG_OK) { std::cerr << "[ICP Error] " << dcreg_last_error(ctx_) << std::endl; return r; }
        if (res.status == 1) std::cerr << "[ICP Warn] Not enough effective points. Aborting." << std::endl;
        if (res.status == 2) std::cerr << "[ICP Error] Solver returned non-finite values!" << std::endl;
        r.converged = res.converged != 0;
        r.time_ms = std::chrono::duration<double, std::milli>(end - start).count();
        r.iterations = res.iterations;
        int n_logged = res.iterations;
        if (res.status == 1 && config_.use_so3_parameterization) n_logged = res.iterations - 1;   // the Euler engine reports iterCount
        n_logged = std::max(0, std::min(n_logged, (int)log.size()));
        r.iteration_data.assign(log.begin(), log.begin() + n_logged);
        for (int i = 0; i < 16; ++i) r.final_transform[i] = (i % 5 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) r.final_transform[i * 4 + j] = res.R[i * 3 + j]; r.final_transform[i * 4 + 3] = res.t[i]; }
        if (!r.iteration_data.empty()) {                                                   // :474-498
            const dcreg_iter_log &last = r.iteration_data.back();
            r.final_rmse = last.rmse; r.final_fitness = last.fitness; r.corr_num = last.effective_points;
            for (int i = 0; i < 16; ++i) r.final_transform[i] = last.transform_matrix[i];
        }
        dcreg_pose_error(config_.gt_matrix, r.final_transform, &r.trans_error_m, &r.rot_error_deg);   // :501-503
        int64_t valid = 0;
        if (dcreg_p2p_error(ctx_, r.final_transform, config_.error_threshold, &r.p2p_rmse, &r.p2p_fitness, &r.chamfer_distance, &valid) == DCREG_OK)
            r.corr_num = valid;                                                            // :508-510
        std::cout << "Translation error: " << r.trans_error_m << " m, Rotation error: " << r.rot_error_deg << " deg" << std::endl;
        std::cout << "P2P RMSE: " << r.p2p_rmse << ", Chamfer: " << r.chamfer_distance << std::endl;
        return r;
    }

    void updateStatistics(const std::string &name, const TestResult &r) {                  // :604-629
        auto &s = statistics_[name];
        s.total_runs++;
        if (r.converged) s.converged_runs++;
        s.mean_trans_error += r.trans_error_m; s.mean_rot_error += r.rot_error_deg; s.mean_time_ms += r.time_ms;
        s.mean_iterations += r.iterations; s.mean_rmse += r.final_rmse; s.mean_fitness += r.final_fitness;
        s.mean_p2p_rmse += r.p2p_rmse; s.mean_p2p_fitness += r.p2p_fitness; s.mean_chamfer += r.chamfer_distance;
        s.corr_num += r.corr_num;
        s.min_trans_error = std::min(s.min_trans_error, r.trans_error_m); s.max_trans_error = std::max(s.max_trans_error, r.trans_error_m);
        s.min_rot_error = std::min(s.min_rot_error, r.rot_error_deg); s.max_rot_error = std::max(s.max_rot_error, r.rot_error_deg);
    }

    void finalizeStatistics() {                                                            // :632-664
        for (auto &kv : statistics_) {
            auto &s = kv.second;
            if (s.total_runs == 0) continue;
            const double n = s.total_runs;
            s.mean_trans_error /= n; s.mean_rot_error /= n; s.mean_time_ms /= n; s.mean_iterations /= n; s.mean_rmse /= n;
            s.mean_fitness /= n; s.mean_p2p_rmse /= n; s.mean_p2p_fitness /= n; s.mean_chamfer /= n;
            s.success_rate = (double)s.converged_runs / n;
            double a = 0, b = 0, c = 0;
            for (const auto &r : detailed_results_[kv.first]) {
                a += std::pow(r.trans_error_m - s.mean_trans_error, 2); b += std::pow(r.rot_error_deg - s.mean_rot_error, 2);
                c += std::pow(r.time_ms - s.mean_time_ms, 2);
            }
            s.std_trans_error = std::sqrt(a / n); s.std_rot_error = std::sqrt(b / n); s.std_time_ms = std::sqrt(c / n);
        }
    }

    void saveStatistics() {                                                                // :667-797
        std::ofstream file(config_.output_folder + "statistics_summary.txt");
        if (!file) { std::cerr << "Failed to open statistics file" << std::endl; return; }
        file << "ICP Test Statistics Summary\n===========================\n\nConfiguration:\n";
        file << "  Source: " << config_.source_pcd << "\n  Target: " << config_.target_pcd << "\n";
        file << "  Cloud size: " << source_.size() << " " << target_.size() << "\n  Runs per method: " << config_.num_runs << "\n\n";
        file << std::fixed << std::setprecision(6);
        file << std::setw(15) << "Method" << std::setw(12) << "Success%" << std::setw(12) << "Trans(m)" << std::setw(12) << "Rot(deg)"
             << std::setw(12) << "ICP_RMSE" << std::setw(12) << "Avg_Iters" << std::setw(12) << "P2PDis" << std::setw(12) << "ChamferDis"
             << std::setw(12) << "P2P_Fit%" << std::setw(12) << "P2P_Corr" << std::setw(12) << "Time(ms)\n";
        file << std::string(135, '-') << "\n";
        for (const auto &kv : statistics_) {
            const auto &s = kv.second;
            file << std::setw(15) << kv.first << std::setw(12) << std::setprecision(1) << (s.success_rate * 100)
                 << std::setw(12) << std::setprecision(4) << s.mean_trans_error << std::setw(12) << s.mean_rot_error << std::setw(12) << s.mean_rmse
                 << std::setw(12) << std::setprecision(1) << s.mean_iterations << std::setw(12) << std::setprecision(4) << s.mean_p2p_rmse
                 << std::setw(12) << s.mean_chamfer << std::setw(12) << std::setprecision(2) << (s.mean_p2p_fitness * 100)
                 << std::setw(12) << s.corr_num << std::setw(12) << std::setprecision(2) << s.mean_time_ms << "\n";
        }
        file << "\n\nDetailed Statistics:\n===================\n\n";
        for (const auto &kv : statistics_) {
            const auto &s = kv.second;
            file << "Method: " << kv.first << "\n";
            file << "  Converged: " << s.converged_runs << "/" << s.total_runs << " (Success Rate: " << std::setprecision(1) << (s.success_rate * 100) << "%)\n";
            file << "  Iterations: " << std::setprecision(1) << s.mean_iterations << "\n";
            file << "  Translation Error (m): " << std::setprecision(6) << s.mean_trans_error << " \xC2\xB1 " << s.std_trans_error << " [" << s.min_trans_error << ", " << s.max_trans_error << "]\n";
            file << "  Rotation Error (deg): " << s.mean_rot_error << " \xC2\xB1 " << s.std_rot_error << " [" << s.min_rot_error << ", " << s.max_rot_error << "]\n";
            file << "  Time (ms): " << std::setprecision(2) << s.mean_time_ms << " \xC2\xB1 " << s.std_time_ms << "\n";
            file << "  ICP RMSE: " << std::setprecision(6) << s.mean_rmse << "\n  ICP Fitness: " << std::setprecision(4) << s.mean_fitness << "\n";
            file << "  ICP Correspondence: " << s.corr_num << "\n";
            file << "  Point-to-Point RMSE: " << std::setprecision(6) << s.mean_p2p_rmse << "\n  Point-to-Point Fitness: " << std::setprecision(4) << s.mean_p2p_fitness << "\n";
            file << "  Chamfer Distance: " << std::setprecision(6) << s.mean_chamfer << "\n\n";
        }
        std::ofstream log(config_.output_folder + "complete_log.txt");
        if (!log) return;
        log << std::fixed << std::setprecision(6);
        log << "Complete ICP Test Log\n====================\n\nConfiguration:\n  Source: " << config_.source_pcd << "\n  Target: " << config_.target_pcd << "\n  Runs: " << config_.num_runs << "\n";
        log << "  Initial noise: x=" << config_.initial_noise.x << ", y=" << config_.initial_noise.y << ", z=" << config_.initial_noise.z
            << ", roll=" << rad2deg(config_.initial_noise.roll) << ", pitch=" << rad2deg(config_.initial_noise.pitch) << ", yaw=" << rad2deg(config_.initial_noise.yaw) << " deg\n\n";
        log << "ICP Parameters:\n  DEGENERACY_THRES_COND: " << config_.core.DEGENERACY_THRES_COND << "\n  DEGENERACY_THRES_EIG: " << config_.core.DEGENERACY_THRES_EIG
            << "\n  STD_REG_GAMMA: " << config_.core.STD_REG_GAMMA << "\n  ADAPTIVE_REG_ALPHA: " << config_.core.ADAPTIVE_REG_ALPHA << "\n  KAPPA_TARGET: " << config_.core.KAPPA_TARGET
            << "\n  PCG_TOLERANCE: " << config_.core.PCG_TOLERANCE << "\n  PCG_MAX_ITER: " << config_.core.PCG_MAX_ITER << "\n\nResults Summary:\n================\n";
        for (const auto &kv : statistics_) {
            const auto &s = kv.second;
            log << "\nMethod: " << kv.first << "\n  Success rate: " << (s.success_rate * 100) << "%\n  Trans error: " << s.mean_trans_error << " \xC2\xB1 " << s.std_trans_error
                << " m\n  Rot error: " << s.mean_rot_error << " \xC2\xB1 " << s.std_rot_error << " deg\n  P2P RMSE: " << s.mean_p2p_rmse << " m\n  Chamfer: " << s.mean_chamfer
                << " m\n  Time: " << s.mean_time_ms << " \xC2\xB1 " << s.std_time_ms << " ms\n";
        }
    }

    void saveDetailedResults() {
        // ---- condition_numbers_detailed.csv (num_runs == 1 only), :893-991 ; default ostream formatting
        if (config_.num_runs == 1) {
            std::ofstream f(config_.output_folder + "condition_numbers_detailed.csv");
            f << "Method,Iteration,Effective_Points,RMSE,Fitness,Cond_Schur_Rot,Cond_Schur_Trans,Cond_Diag_Rot,Cond_Diag_Trans,"
                 "Cond_Full_EVD_Sub_Rot,Cond_Full_EVD_Sub_Trans,Cond_Full_SVD,Lambda_Schur_Rot_0,Lambda_Schur_Rot_1,Lambda_Schur_Rot_2,"
                 "Lambda_Schur_Trans_0,Lambda_Schur_Trans_1,Lambda_Schur_Trans_2,Eigenvalues_Full_0,Eigenvalues_Full_1,Eigenvalues_Full_2,"
                 "Eigenvalues_Full_3,Eigenvalues_Full_4,Eigenvalues_Full_5,Singular_Values_0,Singular_Values_1,Singular_Values_2,"
                 "Singular_Values_3,Singular_Values_4,Singular_Values_5,Is_Degenerate,Degenerate_Mask_0,Degenerate_Mask_1,Degenerate_Mask_2,"
                 "Degenerate_Mask_3,Degenerate_Mask_4,Degenerate_Mask_5\n";
            for (const auto &kv : detailed_results_) {
                if (kv.second.empty()) continue;
                for (const auto &it : kv.second[0].iteration_data) {
                    const dcreg_analysis &a = it.analysis;
                    f << kv.first << "," << it.iter_count << "," << it.effective_points << "," << it.rmse << "," << it.fitness << ",";
                    f << a.cond_schur_rot << "," << a.cond_schur_trans << "," << a.cond_diag_rot << "," << a.cond_diag_trans << ","
                      << a.cond_full_sub_rot << "," << a.cond_full_sub_trans << "," << a.cond_full << ",";
                    for (double v : a.lambda_schur_rot) f << v << ",";
                    for (double v : a.lambda_schur_trans) f << v << ",";
                    for (double v : a.eigenvalues_full) f << v << ",";
                    for (double v : a.singular_values) f << v << ",";
                    f << (a.isDegenerate ? 1 : 0) << ",";
                    for (int i = 0; i < 6; ++i) f << (a.degenerate_mask[i] ? 1 : 0) << (i < 5 ? "," : "");
                    f << "\n";
                }
            }
        }
        // ---- degeneracy_analysis_first_iter.txt (num_runs == 1), :1030-1197
        if (config_.num_runs == 1) {
            std::ofstream f(config_.output_folder + "degeneracy_analysis_first_iter.txt");
            f << "Degeneracy Analysis Results (First Iteration)\n============================================\n\n";
            for (const auto &kv : detailed_results_) {
                if (kv.second.empty()) continue;
                const auto &r = kv.second[0];
                if (r.iteration_data.empty()) { f << "Method: " << kv.first << " - No iteration data available\n\n"; continue; }
                const dcreg_analysis &a = r.iteration_data[0].analysis;
                f << "Method: " << kv.first << "\n  Condition Numbers:\n" << std::fixed << std::setprecision(2);
                f << "    Schur Rot: " << a.cond_schur_rot << "\n    Schur Trans: " << a.cond_schur_trans << "\n    Diag Rot: " << a.cond_diag_rot
                  << "\n    Diag Trans: " << a.cond_diag_trans << "\n    SVD Diag Rot: " << a.cond_full_sub_rot << "\n    SVD Diag Trans: " << a.cond_full_sub_trans
                  << "\n    Full SVD: " << a.cond_full << "\n";
                f << "  Eigenvalues (Full): " << std::setprecision(3);
                for (double v : a.eigenvalues_full) f << v << " ";
                f << "\n  Degenerate Mask (wxwywz xyz): ";
                for (int m : a.degenerate_mask) f << (m ? "1" : "0") << " ";
                f << "\n  Is Degenerate: " << (a.isDegenerate ? "Yes" : "No") << "\n\n" << std::setprecision(6);
                if (kv.first.find("PCG") != std::string::npos || kv.first == "Ours") {
                    // the reference logs P with rows/columns permuted by the alignment indices:
                    // logged(idx[i], idx[j]) = P(i, j) per block; the 3-cycle of the paper run's last iteration
                    // (trans_indices = 1 2 0) fixes the direction, the first iteration's swaps cannot
                    int perm[6];
                    for (int i = 0; i < 3; ++i) { perm[a.rot_indices[i]] = i; perm[3 + a.trans_indices[i]] = 3 + i; }
                    f << "  Preconditioner Matrix P:\n";
                    for (int i = 0; i < 6; ++i) {
                        f << "    ";
                        for (int j = 0; j < 6; ++j) f << std::setw(12) << a.P_preconditioner[perm[i] * 6 + perm[j]] << " ";
                        f << "\n";
                    }
                    f << "\n";
                }
                if ((kv.first == "Ours" || kv.first.find("SCHUR") != std::string::npos) && a.isDegenerate) {
                    f << "  Alignment Analysis:\n";
                    for (int blk = 0; blk < 2; ++blk) {
                        f << (blk ? "    Translation Axes:\n" : "    Rotation Axes:\n");
                        const int *idx = blk ? a.trans_indices : a.rot_indices;
                        const double *lam = blk ? a.lambda_schur_trans : a.lambda_schur_rot;
                        const double *V = blk ? a.aligned_V_trans : a.aligned_V_rot;
                        const char *names = blk ? "XYZ" : "RPY";
                        for (int i = 0; i < 3; ++i) {
                            const double v[3] = {V[0 * 3 + i], V[1 * 3 + i], V[2 * 3 + i]};
                            const double ang = std::acos(std::min(1.0, std::max(0.0, std::fabs(v[i])))) * 180.0 / M_PI;
                            const double sabs = std::max(1e-9, std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]));
                            f << "      [" << i << "]~" << names[i] << " (orig_idx=" << idx[i] << "): \xCE\xBB=" << lam[idx[i]] << ", Angle=" << ang << "\xC2\xB0, "
                              << 100 * std::fabs(v[0]) / sabs << "%" << names[0] << "+" << 100 * std::fabs(v[1]) / sabs << "%" << names[1] << "+"
                              << 100 * std::fabs(v[2]) / sabs << "%" << names[2] << "\n";
                        }
                    }
                    f << " \n";
                }
            }
            f << "\n\n";
        }
        // ---- transform_details.csv, :801-890.  Default ostream formatting.  Bug-compatible: no separator after
        // Transform_33 nor after Degenerate_Mask_5 (header and rows), the "SVD_Sigma" columns carry the eigenvalues, the
        // Schur lambda columns are constant 0.0 and the condition columns hold {Schur rot, Schur trans, full SVD, 0.0, 0.0}.
        {
            std::ofstream f(config_.output_folder + "transform_details.csv");
            f << "Method,Run,Converged,Iterations,Time_ms,Trans_Error_m,Rot_Error_deg,Final_RMSE,Final_Fitness,Corr_Number,";
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) f << "Transform_" << i << j << ",";
            f << "SVD_Sigma_0,SVD_Sigma_1,SVD_Sigma_2,SVD_Sigma_3,SVD_Sigma_4,SVD_Sigma_5,"
                 "EVD_Lambda_0,EVD_Lambda_1,EVD_Lambda_2,EVD_Lambda_3,EVD_Lambda_4,EVD_Lambda_5,"
                 "Schur_Rot_Lambda_0,Schur_Rot_Lambda_1,Schur_Rot_Lambda_2,Schur_Trans_Lambda_0,Schur_Trans_Lambda_1,Schur_Trans_Lambda_2,"
                 "Cond_Full_SVD,Cond_Sub_Rot,Cond_Sub_Trans,Cond_Schur_Rot,Cond_Schur_Trans,"
                 "Degenerate_Mask_0,Degenerate_Mask_1,Degenerate_Mask_2,Degenerate_Mask_3,Degenerate_Mask_4,Degenerate_Mask_5";
            f << "SuperLoc_Has_Data,SuperLoc_Uncertainty_X,SuperLoc_Uncertainty_Y,SuperLoc_Uncertainty_Z,"
                 "SuperLoc_Uncertainty_Roll,SuperLoc_Uncertainty_Pitch,SuperLoc_Uncertainty_Yaw,"
                 "SuperLoc_Cond_Full,SuperLoc_Cond_Rot,SuperLoc_Cond_Trans,SuperLoc_Is_Degenerate\n";
            for (const auto &kv : detailed_results_) {
                int run = 0;
                for (const auto &r : kv.second) {
                    f << kv.first << "," << run++ << "," << (r.converged ? 1 : 0) << "," << r.iterations << "," << r.time_ms << "," << r.trans_error_m << ","
                      << r.rot_error_deg << "," << r.final_rmse << "," << r.final_fitness << "," << r.corr_num << ",";
                    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { f << r.final_transform[i * 4 + j]; if (i < 3 || j < 3) f << ","; }
                    const bool have = !r.iteration_data.empty();                         // result.eigenvalues etc. come from the last iteration, :484-497
                    const dcreg_analysis *a = have ? &r.iteration_data.back().analysis : nullptr;
                    if (have) { for (int rep = 0; rep < 2; ++rep) for (double v : a->eigenvalues_full) f << v << ","; }
                    else for (int i = 0; i < 12; ++i) f << "0.0,";
                    for (int i = 0; i < 6; ++i) f << "0.0,";
                    if (have) f << a->cond_schur_rot << "," << a->cond_schur_trans << "," << a->cond_full << ",0.0,0.0,";
                    else for (int i = 0; i < 5; ++i) f << "0.0,";
                    for (int i = 0; i < 6; ++i) { f << ((have && a->degenerate_mask[i]) ? 1 : 0); if (i < 5) f << ","; }
                    f << "0,NaN,NaN,NaN,NaN,NaN,NaN,NaN,NaN,NaN,0\n";                    // no SuperLoc data in this build
                }
            }
        }
        // ---- degeneracy_analysis_last_iter.txt, :1199-1385 (every num_runs; first run of each method)
        {
            std::ofstream f(config_.output_folder + "degeneracy_analysis_last_iter.txt");
            f << std::fixed << std::setprecision(6);
            f << "Degeneracy Analysis Results\n==========================\n\n";
            auto eigen_row = [](const double *v, int n) {       // Eigen's default IOFormat of a row vector: common width
                std::vector<std::string> t;
                size_t w = 0;
                for (int i = 0; i < n; ++i) { std::ostringstream o; o << std::fixed << std::setprecision(6) << v[i]; t.push_back(o.str()); w = std::max(w, t.back().size()); }
                std::string out;
                for (int i = 0; i < n; ++i) { out += std::string(w - t[i].size(), ' ') + t[i]; if (i + 1 < n) out += " "; }
                return out;
            };
            for (const auto &kv : detailed_results_) {
                if (kv.second.empty()) continue;
                const auto &r = kv.second[0];
                f << "Method: " << kv.first << "\nFinal Transform Matrix:\n";
                for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) f << std::setw(12) << r.final_transform[i * 4 + j] << " "; f << "\n"; }
                f << "\n";
                if (!r.iteration_data.empty()) {
                    const dcreg_analysis &a = r.iteration_data.back().analysis;
                    f << "  Condition Numbers:\n    Schur Rot: " << a.cond_schur_rot << "\n    Schur Trans: " << a.cond_schur_trans << "\n    Diag Rot: " << a.cond_diag_rot
                      << "\n    Diag Trans: " << a.cond_diag_trans << "\n    SVD Diag Rot: " << a.cond_full_sub_rot << "\n    SVD Diag Trans: " << a.cond_full_sub_trans
                      << "\n    Full SVD: " << a.cond_full << "\n\n";
                    f << "  EVD Eigenvalues (Full):\n";
                    for (int i = 0; i < 6; ++i) f << "    \xCE\xBB" << i << ": " << a.eigenvalues_full[i] << "\n";
                    f << "\n  SVD Singular Values:\n";
                    for (int i = 0; i < 6; ++i) f << "    \xCF\x83" << i << ": " << a.singular_values[i] << "\n";
                    f << "\n  Diagonal Block Eigenvalues:\n    Rotation: [" << eigen_row(a.lambda_sub_rot, 3) << "]\n    Translation: [" << eigen_row(a.lambda_sub_trans, 3) << "]\n\n";
                    f << "  Schur Complement Eigenvalues:\n    Rotation: [" << eigen_row(a.lambda_schur_rot, 3) << "]\n    Translation: [" << eigen_row(a.lambda_schur_trans, 3) << "]\n\n";
                    f << "  Degenerate Mask (\xCF\x89x\xCF\x89y\xCF\x89z xyz): ";
                    for (int m : a.degenerate_mask) f << (m ? "1" : "0") << " ";
                    f << "\n\n";
                    if (kv.first.find("PCG") != std::string::npos || kv.first == "Ours") {
                        int perm[6];                                                     // logged(idx[i], idx[j]) = P(i, j), as in the first-iteration file
                        for (int i = 0; i < 3; ++i) { perm[a.rot_indices[i]] = i; perm[3 + a.trans_indices[i]] = 3 + i; }
                        f << "  Preconditioner Matrix P:\n";
                        for (int i = 0; i < 6; ++i) {
                            f << "    ";
                            for (int j = 0; j < 6; ++j) f << std::setw(12) << a.P_preconditioner[perm[i] * 6 + perm[j]] << " ";
                            f << "\n";
                        }
                        f << "\n";
                    }
                    if ((kv.first == "Ours" || kv.first.find("SCHUR") != std::string::npos) && a.isDegenerate) {
                        f << "  Alignment Analysis:\n";
                        for (int blk = 0; blk < 2; ++blk) {
                            f << (blk ? "    Translation Axes:\n" : "    Rotation Axes:\n");
                            const int *idx = blk ? a.trans_indices : a.rot_indices;
                            const double *lam = blk ? a.lambda_schur_trans : a.lambda_schur_rot;
                            const double *V = blk ? a.aligned_V_trans : a.aligned_V_rot;
                            const char *names = blk ? "XYZ" : "RPY";
                            for (int i = 0; i < 3; ++i) {
                                const double v[3] = {V[0 * 3 + i], V[1 * 3 + i], V[2 * 3 + i]};
                                const double ang = std::acos(std::min(1.0, std::max(0.0, std::fabs(v[i])))) * 180.0 / M_PI;
                                const double sabs = std::max(1e-9, std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]));
                                f << "      [" << i << "]~" << names[i] << " (orig_idx=" << idx[i] << "): \xCE\xBB=" << lam[idx[i]] << ", Angle=" << ang << "\xC2\xB0, "
                                  << 100 * std::fabs(v[0]) / sabs << "%" << names[0] << "+" << 100 * std::fabs(v[1]) / sabs << "%" << names[1] << "+"
                                  << 100 * std::fabs(v[2]) / sabs << "%" << names[2] << "\n";
                            }
                        }
                    }
                }
                f << "\n" << std::string(60, '-') << "\n\n";
            }
        }
        // ---- all_results.csv, :996-1028
        {
            std::ofstream f(config_.output_folder + "all_results.csv");
            f << "Method,Run,Converged,Iterations,Time_ms,Trans_Error_m,Rot_Error_deg,ICP_RMSE,ICP_Fitness,P2P_RMSE,P2P_Fitness,Chamfer_Distance\n";
            for (const auto &kv : detailed_results_) {
                int run = 0;
                for (const auto &r : kv.second)
                    f << kv.first << "," << run++ << "," << (r.converged ? 1 : 0) << "," << r.iterations << "," << r.time_ms << "," << r.trans_error_m << ","
                      << r.rot_error_deg << "," << r.final_rmse << "," << r.final_fitness << "," << r.p2p_rmse << "," << r.p2p_fitness << "," << r.chamfer_distance << "\n";
            }
        }
        // ---- iteration_history.csv, :1390-1412
        {
            std::ofstream f(config_.output_folder + "iteration_history.csv");
            f << "Method,Iteration,RMSE,Fitness,TransError,RotError,CorrNum\n" << std::fixed << std::setprecision(8);
            for (const auto &kv : detailed_results_) {
                if (kv.second.empty()) continue;
                for (const auto &it : kv.second[0].iteration_data)
                    f << kv.first << "," << it.iter_count << "," << it.rmse << "," << it.fitness << "," << it.trans_error_vs_gt << "," << it.rot_error_vs_gt << ","
                      << it.effective_points << "\n";
            }
        }
        // ---- iteration_details_with_dx.csv, :1416-1500.  Bug-compatible: the Trans_Error_m / Rot_Error_deg columns are
        // swapped (:1457-1458) and Cond_Sub_* carry cond_diag_*.  P2P columns are recomputed per iteration like the reference.
        {
            std::ofstream f(config_.output_folder + "iteration_details_with_dx.csv");
            f << std::fixed << std::setprecision(8);
            f << "Method,Run,Iteration,RMSE,Fitness,Time_ms,Trans_Error_m,Rot_Error_deg,P2P_RMSE,Chamfer_Distance,dx_wx,dx_wy,dx_wz,dx_x,dx_y,dx_z,"
                 "grad_wx,grad_wy,grad_wz,grad_x,grad_y,grad_z,objective_value,";
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) f << "T_" << i << j << ",";
            f << "Cond_Schur_Rot,Cond_Schur_Trans,Cond_Sub_Rot,Cond_Sub_Trans,Cond_Full_SVD,";
            for (int i = 0; i < 6; ++i) f << "Degenerate_" << i << ",";
            f << "Is_Degenerate\n";
            for (const auto &kv : detailed_results_)
                for (size_t run = 0; run < kv.second.size(); ++run) {
                    const auto &r = kv.second[run];
                    for (size_t k = 0; k < r.iteration_data.size(); ++k) {
                        const auto &it = r.iteration_data[k];
                        double te, re_, p2p = 0, fit = 0, ch = 0; int64_t valid = 0;
                        dcreg_pose_error(config_.gt_matrix, it.transform_matrix, &te, &re_);
                        dcreg_p2p_error(ctx_, it.transform_matrix, config_.error_threshold, &p2p, &fit, &ch, &valid);
                        f << kv.first << "," << run << "," << k << "," << it.rmse << "," << it.fitness << "," << it.iter_time_ms << ",";
                        f << re_ << "," << te << "," << p2p << "," << ch << ",";
                        for (double v : it.update_dx) f << v << ",";
                        for (double v : it.gradient) f << v << ",";
                        f << it.objective_value << ",";
                        for (double v : it.transform_matrix) f << v << ",";
                        const dcreg_analysis &a = it.analysis;
                        f << a.cond_schur_rot << "," << a.cond_schur_trans << "," << a.cond_diag_rot << "," << a.cond_diag_trans << "," << a.cond_full << ",";
                        for (int i = 0; i < 6; ++i) f << (a.degenerate_mask[i] ? 1 : 0) << ",";
                        f << (a.isDegenerate ? 1 : 0) << "\n";
                    }
                }
        }
        std::cout << "Detailed results saved to: " << config_.output_folder << std::endl;
    }
};

}  // namespace

int main(int argc, char **argv) {                                    // icp_main.cpp:6-52 (+ optional config path)
    std::string config_file = "../config/icp.yaml";
    if (argc > 1) config_file = argv[1];
    RunnerConfig config;
    if (!loadConfig(config_file, config)) { std::cerr << "Failed to load configuration from: " << config_file << std::endl; return 1; }
    if (argc > 2) config.output_folder = argv[2];
    if (!config.output_folder.empty() && config.output_folder.back() != '/') config.output_folder += "/";
    fs::create_directories(config.output_folder);
    TestRunner runner(config);
    std::cout << "\n========================================\nStarting ICP Test Suite\nNumber of methods: " << config.test_methods.size()
              << "\nNumber of runs per method: " << config.num_runs << "\n========================================\n" << std::endl;
    const auto start = std::chrono::high_resolution_clock::now();
    if (!runner.runAllTests()) { std::cerr << "Test execution failed!" << std::endl; return 1; }
    const auto secs = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::high_resolution_clock::now() - start).count();
    std::cout << "\n========================================\nAll tests completed successfully!\nTotal time: " << secs << " seconds\nResults saved to: "
              << config.output_folder << "\n========================================" << std::endl;
    return 0;
}
