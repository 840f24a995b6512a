// probabilistic_point_cloud_registration — command-line front end.  Flags, defaults, printed lines and output files
// are the reference's (src/prob_point_cloud_registration_ex.cc:34-66,93-188, README.md:31-92):
//   probabilistic_point_cloud_registration [--dump] [-g gt.pcd] [-v] [-u] [-n int] [-c float] [-r float]
//       [-d float] [-i int] [-m int] [-t float] [-s float] <source.pcd> <target.pcd>
// Additions (they do not collide with the reference's letters): --device N, --inner-steps K, and the batch front end
//   probabilistic_point_cloud_registration --batch pairs.txt [--lanes L] [-r -m -d -u -i -c -n --inner-steps as above]
// where every non-empty line of pairs.txt names "<source.pcd> <target.pcd>": the pairs are registered by ppcr_batch_run
// over every visible GPU (pair p on device p % n, L pairs in flight per device) and one line per pair is printed.
// The program is organised as a small pipeline of its own: parse -> load -> register -> publish.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "prob_point_cloud_registration/adapters.hpp"
#include "prob_point_cloud_registration/pcd_io.hpp"
#include "prob_point_cloud_registration/prob_point_cloud_registration.h"
#include "prob_point_cloud_registration/utilities.hpp"

namespace ppcr_cli {

namespace fs = std::filesystem;
namespace reg = prob_point_cloud_registration;
using Cloud = pcl::PointCloud<pcl::PointXYZ>;

struct Job {
    reg::ProbPointCloudRegistrationParams params;
    std::string source_path, target_path, truth_path;  // truth_path empty: no -g
    std::string batch_path;                            // --batch: a list of pairs instead of one pair
    int lanes = 2;                                     // --lanes: pairs in flight per device (batch mode)
    // one process per GPU (batch mode): this process is rank `rank` of `world`, registers the pairs p with p % world == rank
    // on its device and takes part in the final RCCL all-gather of the transforms; rank 0 prints them
    int rank = 0, world = 1;
    std::string rendezvous;                            // --rendezvous: a file rank 0 leaves the communicator id in
    std::string run_id;                                // --run-id: a token of THIS launch; ranks only accept a rendezvous file that carries it
    bool device_given = false;                         // --device on the command line (rank mode: default rank % devices)
    bool gaussian = false;
};

struct BadArgument {
    std::string complaint, argument;
};

[[noreturn]] void rejectCommandLine(const BadArgument &bad)
{
    std::cerr << "error: " << bad.complaint << " for arg " << bad.argument << std::endl;
    std::cerr << "usage: probabilistic_point_cloud_registration [--dump] [-g <string>] [-v] [-u] [-n <int>] [-c <float>]\n"
                 "         [-r <float>] [-d <float>] [-i <int>] [-m <int>] [-t <float>] [-s <float>] [--device <int>]\n"
                 "         [--inner-steps <int>] <source_file_name> <target_file_name>\n"
                 "   or: probabilistic_point_cloud_registration --batch <pair_list_file> [--lanes <int>] [options as above]\n"
                 "         [--rank <int> --world <int> --rendezvous <file> [--run-id <string>]]   (one process per GPU, RCCL gather of the transforms)"
              << std::endl;
    std::exit(EXIT_FAILURE);
}

// cursor over argv with typed reads; every failure names the flag it happened on
class ArgCursor {
public:
    ArgCursor(int argc, char **argv) : args_(argv + 1, argv + argc) {}
    bool done() const { return next_ >= args_.size(); }
    std::string take() { return args_[next_++]; }
    std::string valueOf(const std::string &flag)
    {
        if (done()) throw BadArgument{"Missing a value", flag};
        return take();
    }
    template <class T, class Parse>
    T number(const std::string &flag, Parse parse)
    {
        const std::string text = valueOf(flag);
        std::size_t used = 0;
        T v{};
        try {
            v = parse(text, &used);
        } catch (...) {
            used = std::string::npos;
        }
        if (used != text.size()) throw BadArgument{"Couldn't read argument value from string '" + text + "'", flag};
        return v;
    }
    int integer(const std::string &flag)
    {
        return number<int>(flag, [](const std::string &s, std::size_t *n) { return std::stoi(s, n); });
    }
    float real(const std::string &flag)
    {
        return number<float>(flag, [](const std::string &s, std::size_t *n) { return std::stof(s, n); });
    }

private:
    std::vector<std::string> args_;
    std::size_t next_ = 0;
};

bool looksLikeFlag(const std::string &a)
{
    // "-3" and "-.5" are values, not flags
    return a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9') && a[1] != '.';
}

Job parseCommandLine(int argc, char **argv)
{
    Job job;
    // the CLI's own defaults (radius 3 here, 1 in the params struct — the reference has the same split)
    job.params.radius = 3;
    job.params.max_neighbours = 20;
    job.params.n_iter = 1000;
    job.params.dof = 5;
    job.params.cost_drop_thresh = 0.01;
    job.params.n_cost_drop_it = 5;
    std::vector<std::string> files;
    ArgCursor cur(argc, argv);
    while (!cur.done()) {
        const std::string a = cur.take();
        auto is = [&](const char *shortf, const char *longf) { return a == shortf || a == longf; };
        if (is("-r", "--radius")) job.params.radius = cur.real(a);
        else if (is("-m", "--max_neighbours")) job.params.max_neighbours = cur.integer(a);
        else if (is("-i", "--num_iter")) job.params.n_iter = cur.integer(a);
        else if (is("-d", "--dof")) job.params.dof = cur.real(a);
        else if (is("-c", "--cost_drop_treshold")) job.params.cost_drop_thresh = cur.real(a);
        else if (is("-n", "--num_drop_iter")) job.params.n_cost_drop_it = cur.integer(a);
        else if (is("-s", "--source_filter_size")) job.params.source_filter_size = cur.real(a);
        else if (is("-t", "--target_filter_size")) job.params.target_filter_size = cur.real(a);
        else if (is("-g", "--ground_truth")) job.truth_path = cur.valueOf(a);
        else if (is("-u", "--use_gaussian")) job.gaussian = true;
        else if (is("-v", "--verbose")) job.params.verbose = true;
        else if (a == "--dump") job.params.summary = true;
        else if (a == "--device") job.params.device_id = cur.integer(a), job.device_given = true;
        else if (a == "--inner-steps") job.params.inner_max_steps = cur.integer(a);
        else if (a == "--batch") job.batch_path = cur.valueOf(a);
        else if (a == "--lanes") job.lanes = cur.integer(a);
        else if (a == "--rank") job.rank = cur.integer(a);
        else if (a == "--world") job.world = cur.integer(a);
        else if (a == "--rendezvous") job.rendezvous = cur.valueOf(a);
        else if (a == "--run-id") job.run_id = cur.valueOf(a);
        else if (is("-h", "--help")) throw BadArgument{"help requested", a};
        else if (looksLikeFlag(a)) throw BadArgument{"Couldn't find match for argument", a};
        else files.push_back(a);
    }
    static const char *const kNames[2] = {"source_file_name", "target_file_name"};
    if (!job.batch_path.empty()) {
        if (!files.empty()) throw BadArgument{"Positional arguments are not used with --batch", files[0]};
        if (job.lanes < 1) throw BadArgument{"--lanes must be at least 1", "--lanes"};
        if (job.world < 1 || job.rank < 0 || job.rank >= job.world) throw BadArgument{"--rank must lie in [0, --world)", "--rank"};
        if (job.world > 1 && job.rendezvous.empty()) throw BadArgument{"--world > 1 needs --rendezvous <file>", "--world"};
        // the batch entry point takes clouds as they are and reports transforms only: refuse what it would silently drop
        if (job.params.source_filter_size > 0) throw BadArgument{"Voxel filters belong to the single-pair form, not --batch", "-s"};
        if (job.params.target_filter_size > 0) throw BadArgument{"Voxel filters belong to the single-pair form, not --batch", "-t"};
        if (!job.truth_path.empty()) throw BadArgument{"The ground-truth report belongs to the single-pair form, not --batch", "-g"};
        if (job.params.summary) throw BadArgument{"The registration report belongs to the single-pair form, not --batch", "--dump"};
    } else {
        if (files.size() < 2) throw BadArgument{"Required argument missing", kNames[files.size()]};
        if (files.size() > 2) throw BadArgument{"Too many positional arguments", files[2]};
        job.source_path = files[0];
        job.target_path = files[1];
    }
    if (job.gaussian) job.params.dof = std::numeric_limits<double>::infinity();  // -u: Gaussian weights
    return job;
}

void announce(const Job &job)
{
    if (!job.params.verbose) return;
    const auto &p = job.params;
    if (job.gaussian) std::cout << "Using gaussian model" << std::endl;
    else std::cout << "Using a t-distribution with " << p.dof << " dof" << std::endl;
    std::cout << "Radius of the neighborhood search: " << p.radius << std::endl;
    std::cout << "Max number of neighbours: " << p.max_neighbours << std::endl;
    std::cout << "Max number of iterations: " << p.n_iter << std::endl;
    std::cout << "Cost drop threshold: " << p.cost_drop_thresh << std::endl;
    std::cout << "Num cost drop iter: " << p.n_cost_drop_it << std::endl;
}

// nullptr when the file cannot be read (the reader has already said why on stderr)
Cloud::Ptr readCloud(const std::string &path)
{
    auto cloud = std::make_shared<Cloud>();
    if (reg::io::loadPCDFile(path, *cloud) == -1) return nullptr;
    return cloud;
}

Cloud::Ptr readCloudOrExit(const std::string &path, const char *role, bool verbose)
{
    if (verbose) std::cout << "Loading " << role << " point cloud from " << path << std::endl;
    Cloud::Ptr cloud = readCloud(path);
    if (!cloud) {
        std::cout << "Could not load " << role << " cloud, closing" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    return cloud;
}

void printHistory(const std::vector<Eigen::Affine3d> &history)
{
    std::cout << "Transformation history:" << std::endl;
    for (const Eigen::Affine3d &T : history) {
        const Eigen::Quaterniond q(T.rotation());
        std::cout << "T: " << T.translation().x() << ", " << T.translation().y() << ", " << T.translation().z()
                  << " ||| R: " << q.x() << ", " << q.y() << ", " << q.z() << ", " << q.w() << std::endl;
    }
}

void saveAligned(const Job &job, const Cloud &aligned)
{
    const std::string name = "aligned_" + fs::path(job.source_path).filename().string();
    std::cout << "Saving aligned source cloud to: " << name << std::endl;
    reg::io::savePCDFile(name, aligned);
}

void writeSummary(const Job &job, const std::string &table)
{
    const std::string name = fs::path(job.source_path).stem().string() + "_" + fs::path(job.target_path).stem().string() +
                             "_summary.txt";
    std::cout << "Saving registration report to: " << name << std::endl;
    const auto &p = job.params;
    std::ofstream out(name);
    out << "Source: " << job.source_path << " with filter size: " << p.source_filter_size << std::endl;
    out << "Target:" << job.target_path << " with filter size: " << p.target_filter_size << std::endl;
    out << "dof: " << p.dof << " | Radius: " << p.radius << " | Max_iter: " << p.n_iter << " | Max neigh: " << p.max_neighbours
        << " | Cost_drop_thresh_: " << p.cost_drop_thresh << " | N_cost_drop_it: " << p.n_cost_drop_it << std::endl;
    out << table;
}

// one-process-per-GPU mode of --batch: the communicator of the final gather, made BEFORE any cloud is read (every rank
// is then at the same point of its life, and a rank that fails later still takes part in the collectives: runBatch).
// Rank 0 removes whatever an earlier launch left under the rendezvous name, draws the RCCL id and leaves a record
// {magic, time written, --run-id token, id} there (written under another name and renamed, so a reader never sees half
// of it).  The other ranks wait for a record of THIS launch: one that carries their --run-id token when a token was
// given, else one written no earlier than a minute before they started (a file left by a crashed launch is older, or
// carries another token).  ppcr_comm_create is collective.
struct RendezvousRecord {
    char magic[8];
    std::int64_t written_unix_ms;
    char run_id[64];
    unsigned char id[PPCR_COMM_ID_BYTES];
};
constexpr char kRendezvousMagic[8] = {'P', 'P', 'C', 'R', 'r', 'v', '0', '2'};

std::int64_t unixMillis()
{
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

ppcr_comm *joinRanks(const Job &job, int device, std::int64_t started_unix_ms)
{
    RendezvousRecord rec;
    std::memset(&rec, 0, sizeof rec);
    if (job.rank == 0) {
        std::error_code ec;
        fs::remove(job.rendezvous, ec);  // (a crashed launch's record: nobody of this launch may take it for ours)
        if (ppcr_comm_get_id(rec.id) != PPCR_OK) return nullptr;
        std::memcpy(rec.magic, kRendezvousMagic, sizeof rec.magic);
        rec.written_unix_ms = unixMillis();
        std::snprintf(rec.run_id, sizeof rec.run_id, "%s", job.run_id.c_str());
        const std::string tmp = job.rendezvous + ".tmp";
        {
            std::ofstream out(tmp, std::ios::binary);
            out.write(reinterpret_cast<const char *>(&rec), sizeof rec);
            if (!out) return nullptr;
        }
        fs::rename(tmp, job.rendezvous, ec);
        if (ec) return nullptr;
    } else {
        bool have = false;
        for (int tries = 0; tries < 1200 && !have; tries++) {  // two minutes from this rank's start
            std::error_code ec;
            if (fs::exists(job.rendezvous, ec) && fs::file_size(job.rendezvous, ec) == sizeof rec) {
                std::ifstream in(job.rendezvous, std::ios::binary);
                in.read(reinterpret_cast<char *>(&rec), sizeof rec);
                have = in.gcount() == static_cast<std::streamsize>(sizeof rec) && std::memcmp(rec.magic, kRendezvousMagic, sizeof rec.magic) == 0;
                rec.run_id[sizeof rec.run_id - 1] = 0;
                if (have && !job.run_id.empty()) {
                    have = job.run_id.compare(0, sizeof rec.run_id - 1, rec.run_id) == 0;
                } else if (have) {
                    // No --run-id: only a record written around this rank's own start can be this launch's (ranks of one
                    // launch start within seconds of each other) — and only one that is still there, unchanged, half a second
                    // later: rank 0 of THIS launch removes and rewrites the file first thing, so a valid record a crashed
                    // launch left moments ago changes under a rank that read it too early.  (--run-id is the safe way.)
                    have = rec.written_unix_ms >= started_unix_ms - 10000;
                    if (have) {
                        std::this_thread::sleep_for(std::chrono::milliseconds(500));
                        RendezvousRecord again;
                        std::memset(&again, 0, sizeof again);
                        std::ifstream in2(job.rendezvous, std::ios::binary);
                        in2.read(reinterpret_cast<char *>(&again), sizeof again);
                        have = in2.gcount() == static_cast<std::streamsize>(sizeof again) && std::memcmp(&again, &rec, sizeof rec) == 0;
                    }
                }
            }
            if (!have) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!have) {
            std::cerr << "rank " << job.rank << ": no communicator id of this launch appeared in " << job.rendezvous << std::endl;
            return nullptr;
        }
    }
    ppcr_comm *comm = nullptr;
    if (ppcr_comm_create(device, job.rank, job.world, rec.id, &comm) != PPCR_OK) return nullptr;
    return comm;
}

// --batch: every pair of the list through ppcr_batch_run — on all visible devices of this process, or (with --rank /
// --world / --rendezvous: one process per GPU) this rank's share on its device, then the RCCL gather of the transforms
int runBatch(const Job &job)
{
    const std::int64_t started = unixMillis();
    std::ifstream list(job.batch_path);
    if (!list) {
        std::cout << "Could not read the pair list " << job.batch_path << ", closing" << std::endl;
        return EXIT_FAILURE;
    }
    const bool ranked = job.world > 1 || !job.rendezvous.empty();
    std::vector<std::pair<std::string, std::string>> files;
    std::string line;
    while (std::getline(list, line)) {
        std::istringstream fields(line);
        std::string src_name, tgt_name;
        if (!(fields >> src_name)) continue;  // blank line
        if (src_name[0] == '#') continue;
        if (!(fields >> tgt_name)) {
            std::cout << "Pair list line without a target: " << line << std::endl;
            return EXIT_FAILURE;
        }
        files.emplace_back(src_name, tgt_name);
    }
    const std::size_t n_pairs = files.size();
    const auto &p = job.params;
    int n_devices = 0;
    ppcr_device_count(&n_devices);
    if (n_devices < 1) {
        std::cerr << "registration failed: no HIP device visible (this library has no CPU fallback)" << std::endl;
        return EXIT_FAILURE;
    }
    std::vector<int> devices;
    if (ranked) devices.push_back(job.device_given ? p.device_id : job.rank % n_devices);
    else
        for (int d = 0; d < n_devices; d++) devices.push_back(d);
    // Ranked: the communicator FIRST.  Every rank reads the same list and arrives here within its start-up time; from here
    // on a rank that fails (a cloud it cannot read, a registration error) keeps its place in the collectives and reports
    // the failure through them, so that no rank is left waiting inside one.
    ppcr_comm *comm = nullptr;
    if (ranked) {
        comm = joinRanks(job, devices[0], started);
        if (!comm) {
            std::cerr << "rank " << job.rank << ": could not join the ranks: " << ppcr_comm_last_error() << std::endl;
            return EXIT_FAILURE;
        }
    }
    // the pairs THIS process registers: all of them, or p % world == rank (the deal of ppcr_batch_run and batch.py)
    std::vector<std::size_t> mine;
    for (std::size_t k = 0; k < n_pairs; k++)
        if (!ranked || static_cast<int>(k % static_cast<std::size_t>(job.world)) == job.rank) mine.push_back(k);
    bool local_ok = true;
    std::vector<Cloud::Ptr> clouds;  // source, target, source, target, ... of `mine`
    for (std::size_t k : mine) {
        for (const std::string *name : {&files[k].first, &files[k].second}) {
            Cloud::Ptr cloud = readCloud(*name);
            if (!cloud) {
                std::cout << "Could not load " << *name << ", closing" << std::endl;
                local_ok = false;
                break;
            }
            clouds.push_back(cloud);
        }
        if (!local_ok) break;
    }
    std::vector<double> T_mine(12 * mine.size(), std::numeric_limits<double>::quiet_NaN());
    std::vector<int32_t> it_mine(mine.size(), 0);
    if (local_ok) {
        std::vector<ppcr_pair> pairs(mine.size());
        for (std::size_t k = 0; k < mine.size(); k++) {
            const Cloud &s = *clouds[2 * k], &t = *clouds[2 * k + 1];
            pairs[k] = ppcr_pair{s.empty() ? nullptr : &s[0].x, static_cast<int64_t>(s.size()), sizeof(pcl::PointXYZ),
                                 t.empty() ? nullptr : &t[0].x, static_cast<int64_t>(t.size()), sizeof(pcl::PointXYZ)};
        }
        ppcr_batch_options opt{};
        opt.radius = p.radius;
        opt.dof = p.dof;
        opt.cost_drop_thresh = p.cost_drop_thresh;
        opt.n_cost_drop_it = p.n_cost_drop_it;
        opt.f_tol = 10e-6;  // the reference's function_tolerance
        for (int k = 0; k < 4; k++) opt.q0[k] = p.initial_rotation[k];
        for (int k = 0; k < 3; k++) opt.t0[k] = p.initial_translation[k];
        opt.max_neighbours = p.max_neighbours;
        opt.dim = 3;
        opt.n_iter = p.n_iter;
        opt.inner_steps = p.inner_max_steps;
        char err[512] = {0};
        if (job.params.verbose)
            std::cout << "Registering " << mine.size() << " of " << n_pairs << " pairs on " << devices.size() << " device(s), " << job.lanes
                      << " in flight each" << (ranked ? " (rank " + std::to_string(job.rank) + " of " + std::to_string(job.world) + ")" : std::string())
                      << std::endl;
        const int rc = ppcr_batch_run(pairs.data(), static_cast<int64_t>(pairs.size()), &opt, devices.data(), static_cast<int>(devices.size()),
                                      job.lanes, T_mine.data(), it_mine.data(), err, sizeof err);
        if (rc != PPCR_OK) {
            std::cerr << "registration failed: " << err << std::endl;
            local_ok = false;
        }
    }
    if (!ranked) {
        if (!local_ok) return EXIT_FAILURE;
    }
    std::vector<double> T(12 * n_pairs);
    std::vector<int32_t> iterations(n_pairs);
    if (ranked) {
        // the job's collectives: every rank ends up with every transform, (a second record) every iteration count and (a
        // third, one slot per rank) every rank's verdict — a failed rank sends NaN transforms and says so
        std::vector<double> counts_mine(12 * mine.size(), 0.0), counts(12 * n_pairs, 0.0);
        for (std::size_t k = 0; k < mine.size(); k++) counts_mine[12 * k] = it_mine[k];
        const std::size_t world = static_cast<std::size_t>(job.world);
        std::vector<double> verdict_mine(12, 0.0), verdicts(12 * world, 0.0);
        verdict_mine[0] = local_ok ? 1.0 : 0.0;
        bool ok = ppcr_gather_transforms(comm, verdict_mine.data(), static_cast<int64_t>(world), verdicts.data()) == PPCR_OK &&
                  ppcr_gather_transforms(comm, T_mine.data(), static_cast<int64_t>(n_pairs), T.data()) == PPCR_OK &&
                  ppcr_gather_transforms(comm, counts_mine.data(), static_cast<int64_t>(n_pairs), counts.data()) == PPCR_OK;
        if (!ok) std::cerr << "gather of the transforms failed: " << ppcr_comm_last_error() << std::endl;
        ppcr_comm_destroy(comm);
        if (job.rank == 0) {
            std::error_code ec;
            fs::remove(job.rendezvous, ec);
        }
        if (!ok) return EXIT_FAILURE;
        for (std::size_t r = 0; r < world; r++)
            if (!(verdicts[12 * r] == 1.0)) {
                if (job.rank == 0 || static_cast<int>(r) == job.rank) std::cerr << "rank " << r << " failed: no result is reported" << std::endl;
                ok = false;
            }
        if (!ok) return EXIT_FAILURE;
        for (std::size_t k = 0; k < n_pairs; k++) iterations[k] = static_cast<int32_t>(counts[12 * k]);
        if (job.rank != 0) return EXIT_SUCCESS;  // rank 0 reports
    } else {
        T = T_mine;
        iterations = it_mine;
    }
    for (std::size_t k = 0; k < n_pairs; k++) {
        const Eigen::Affine3d A = reg::affineFromRows(&T[12 * k]);
        const Eigen::Quaterniond q(A.rotation());
        std::cout << "pair " << k << " (" << files[k].first << " -> " << files[k].second << "), " << iterations[k]
                  << " iterations: T: " << A.translation().x() << ", " << A.translation().y() << ", " << A.translation().z()
                  << " ||| R: " << q.x() << ", " << q.y() << ", " << q.z() << ", " << q.w() << std::endl;
    }
    return EXIT_SUCCESS;
}

int run(const Job &job)
{
    if (!job.batch_path.empty()) {
        announce(job);
        return runBatch(job);
    }
    announce(job);
    const bool verbose = job.params.verbose;
    const Cloud::Ptr source = readCloudOrExit(job.source_path, "source", verbose);
    const Cloud::Ptr target = readCloudOrExit(job.target_path, "target", verbose);
    Cloud::Ptr truth;
    if (!job.truth_path.empty()) {
        std::cout << "Loading ground truth point cloud from " << job.truth_path << std::endl;
        truth = readCloud(job.truth_path);
        if (!truth) std::cout << "Could not load ground truth" << std::endl;  // the run goes on without it
    }

    std::unique_ptr<reg::ProbPointCloudRegistration> solver;
    try {
        if (truth) solver = std::make_unique<reg::ProbPointCloudRegistration>(source, target, job.params, truth);
        else solver = std::make_unique<reg::ProbPointCloudRegistration>(source, target, job.params);
        if (verbose) std::cout << "Registration\n";
        solver->align();
    } catch (const std::exception &e) {
        std::cerr << "registration failed: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    const std::vector<Eigen::Affine3d> history = solver->transformation_history();
    if (history.empty()) {
        std::cerr << "no iteration was performed (num_iter = 0?)" << std::endl;
        return EXIT_FAILURE;
    }

    // the final transform is applied to the cloud as it was read from disk
    auto aligned = std::make_shared<Cloud>();
    pcl::transformPointCloud(*source, *aligned, history.back());
    if (verbose) {  // the aligned cloud is only saved in verbose mode (a quirk of the reference, kept)
        printHistory(history);
        saveAligned(job, *aligned);
    }
    if (job.params.summary) writeSummary(job, solver->report());
    if (truth) std::cout << "MSE w.r.t. ground truth: " << reg::calculateMSE(aligned, truth) << std::endl;
    return EXIT_SUCCESS;
}

}  // namespace ppcr_cli

int main(int argc, char **argv)
{
    try {
        return ppcr_cli::run(ppcr_cli::parseCommandLine(argc, argv));
    } catch (const ppcr_cli::BadArgument &bad) {
        ppcr_cli::rejectCommandLine(bad);
    }
}
