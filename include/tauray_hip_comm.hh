// tauray_hip_comm.hh - rt_renderer for one process per GPU: every process owns one device and one share of the frame, the
// partial frames meet on rank 0 through RCCL (include/trhip_comm.h; link with -ltrhip -ltrhip_comm).
//
// The reference drives all devices from one process and moves partial frames with device_transfer: GPU -> pinned host memory
// -> GPU, paced by timeline semaphores exported between the devices (src/device_transfer.cc:21-347, created per device pair by
// src/rt_renderer.cc:356-408).  tr::rt_renderer of tauray_hip.hh keeps that arrangement with peer copies; this header is the
// other arrangement north_star names - one rank per GPU, "per-device partial framebuffers reduced via RCCL over xGMI before
// tonemap".  The render() sequence is the reference's (src/rt_renderer.cc:84-133): ray tracer -> transfer -> stitch -> tonemap,
// enqueued on the frame slot's stream without blocking the host; ranks other than 0 stop after the transfer.
#pragma once
#include "tauray_hip.hh"
#include "trhip_comm.h"

#include <chrono>
#include <cstring>
#include <ctime>
#include <fstream>
#include <thread>
#include <sys/stat.h>

namespace tr
{

inline void check_comm(int rc)
{
    if(rc != 0) throw std::runtime_error(trhip_comm_last_error());
}

// The 128-byte RCCL id has to reach every rank by the caller's means; for processes that share a file system: rank 0 writes
// it to `path` (atomically, through a rename), the others wait for the file.  A file left behind by an earlier or crashed job
// must not be taken for this job's: the file starts with a 16-byte header - "TRHIPCID" + a 64-bit nonce the launcher hands to
// every rank (`--comm-nonce`, e.g. its pid or start time) - and a reader only accepts a file that carries its own nonce.  Rank 0
// removes whatever is at `path` before it creates the id, and removes the file again once the communicator exists on every rank
// (remove_comm_id_file after trhip_comm_create returns, which is collective).  Without a nonce (0) readers fall back to the
// age of the file: one written more than `stale_seconds` before the reader started belongs to another job.
inline std::vector<char> exchange_comm_id_through_file(const std::string& path, int rank, uint64_t nonce = 0, double timeout_seconds = 120.0,
                                                       double stale_seconds = 60.0)
{
    static const char magic[8] = {'T', 'R', 'H', 'I', 'P', 'C', 'I', 'D'};
    std::vector<char> id(TRHIP_COMM_ID_BYTES);
    if(rank == 0)
    {
        std::remove(path.c_str());      // a previous job's file: no reader of this job may see it once our id exists
        check_comm(trhip_comm_unique_id(id.data()));
        const std::string tmp = path + ".tmp";
        {
            std::ofstream f(tmp, std::ios::binary);
            f.write(magic, 8); f.write(reinterpret_cast<const char*>(&nonce), 8); f.write(id.data(), (std::streamsize)id.size());
            if(!f) throw std::runtime_error("cannot write " + tmp);
        }
        if(std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
        return id;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const std::time_t wall0 = std::time(nullptr);
    while(true)
    {
        std::ifstream f(path, std::ios::binary);
        char head[16];
        if(f && f.read(head, 16) && std::memcmp(head, magic, 8) == 0 && f.read(id.data(), (std::streamsize)id.size()))
        {
            uint64_t file_nonce; std::memcpy(&file_nonce, head + 8, 8);
            bool ours = file_nonce == nonce;
            if(ours && nonce == 0)
            {   // no nonce to tell jobs apart: a file much older than this process is a leftover
                struct stat st;
                if(::stat(path.c_str(), &st) == 0) ours = std::difftime(wall0, st.st_mtime) <= stale_seconds;
            }
            if(ours) return id;
        }
        if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds)
            throw std::runtime_error("timed out waiting for the communicator id of this job in " + path);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}
// rank 0, after trhip_comm_create has returned (every rank has read the id by then: the call is collective)
inline void remove_comm_id_file(const std::string& path, int rank) { if(rank == 0) std::remove(path.c_str()); }

// The set-up of the copy-engine exchange (trhip_ipc_*) needs every rank's 256-byte blob on every rank: an all-gather through files
// `prefix.ipc<rank>` for processes that share a file system, each with the job's nonce in front (as the communicator id above: a file of
// another job is waited out).  The nonce is required here: a blob holds a hipIpcMemHandle, and with nothing to tell jobs apart a rank
// of a second job on the same prefix could read a peer's blob of the first job before that peer has replaced it, and map a dead or foreign
// allocation.  A second round removes the files: every rank leaves a marker `prefix<suffix><rank>.read` once it has read all blobs, and removes
// its own blob when every rank's marker of this job is there (the markers - nonce and nothing else - stay until the next job replaces them).
inline std::vector<char> allgather_blobs_through_files(const std::string& prefix, int rank, int nranks, const std::vector<char>& blob, uint64_t nonce,
                                                       double timeout_seconds = 120.0, const std::string& suffix = ".ipc")
{
    if(nonce == 0) throw std::runtime_error("allgather_blobs_through_files: needs a non-zero nonce shared by the ranks of this job (--comm-nonce)");
    auto path_of = [&](int r) { return prefix + suffix + std::to_string(r); };
    auto marker_of = [&](int r) { return prefix + suffix + std::to_string(r) + ".read"; };
    std::remove(marker_of(rank).c_str());
    {
        const std::string tmp = path_of(rank) + ".tmp";
        std::remove(path_of(rank).c_str());
        { std::ofstream f(tmp, std::ios::binary); f.write(reinterpret_cast<const char*>(&nonce), 8); f.write(blob.data(), (std::streamsize)blob.size()); if(!f) throw std::runtime_error("cannot write " + tmp); }
        if(std::rename(tmp.c_str(), path_of(rank).c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
    }
    std::vector<char> all((size_t)nranks * blob.size());
    const auto t0 = std::chrono::steady_clock::now();
    for(int r = 0; r < nranks; ++r)
    {
        while(true)
        {
            std::ifstream f(path_of(r), std::ios::binary);
            uint64_t file_nonce = 0;
            if(f && f.read(reinterpret_cast<char*>(&file_nonce), 8) && f.read(all.data() + (size_t)r * blob.size(), (std::streamsize)blob.size()) && file_nonce == nonce)
                break;
            if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds)
                throw std::runtime_error("timed out waiting for rank " + std::to_string(r) + " in " + path_of(r));
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    {   // second round: this rank has read everything; once everybody has, the blobs have done their job
        const std::string tmp = marker_of(rank) + ".tmp";
        { std::ofstream f(tmp, std::ios::binary); f.write(reinterpret_cast<const char*>(&nonce), 8); if(!f) throw std::runtime_error("cannot write " + tmp); }
        if(std::rename(tmp.c_str(), marker_of(rank).c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
        for(int r = 0; r < nranks; ++r)
        {
            while(true)
            {
                std::ifstream f(marker_of(r), std::ios::binary);
                uint64_t file_nonce = 0;
                if(f && f.read(reinterpret_cast<char*>(&file_nonce), 8) && file_nonce == nonce) break;
                if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds)
                    throw std::runtime_error("timed out waiting for rank " + std::to_string(r) + " to read the blobs (" + marker_of(r) + ")");
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        }
        std::remove(path_of(rank).c_str());
    }
    return all;
}

template<typename Pipeline>
class basic_process_rt_renderer
{
public:
    using options = typename basic_rt_renderer<Pipeline>::options;

    // `rank` of `nranks` processes, this one on HIP device `hip_device`; `comm_id`: TRHIP_COMM_ID_BYTES bytes every rank holds (RCCL
    // carries the partial frames), or nullptr: no RCCL communicator - use_copy_engine_exchange() before the first frame.
    basic_process_rt_renderer(int hip_device, int rank, int nranks, const void* comm_id, const scene_data& scene, uvec2 size, options opt)
    : rank(rank), nranks(nranks), size(size), opt(opt), dev(hip_device), scene_update(dev), hip_device(hip_device)
    {
        if(nranks < 1 || rank < 0 || rank >= nranks) throw std::runtime_error("process_rt_renderer: rank out of range");
        if(nranks == 1) this->opt.distribution.strategy = DISTRIBUTION_DUPLICATE;   // src/tauray.cc:519-521
        const int n_slots = std::max(this->opt.max_frames_in_flight, 1);
        if(n_slots > 1 && this->opt.accumulate)
            throw std::runtime_error("process_rt_renderer: accumulating frames depend on each other, frames in flight must be 1");
        if(comm_id) check_comm(trhip_comm_create(hip_device, nranks, rank, comm_id, &comm));
        scene_update.set_scene(scene);                                // the scene is replicated on every device (src/gpu_buffer.hh:63-116)
        layers = this->opt.active_viewport_count;
        set_dists(std::vector<double>((size_t)nranks, 1.0 / nranks));
        slots.resize((size_t)n_slots);
        for(slot_data& sl: slots)
        {
            sl.stream = dev.create_stream();
            // targets have the size of the largest share set_device_workloads can hand a rank (src/rt_renderer.cc init_resources)
            const uvec2 ms = get_distribution_target_max_size(dists[(size_t)rank]);
            sl.color = dev.alloc(size_t(ms.x) * ms.y * 16 * layers);
            check(trhip_memset(dev.h, sl.color, 0, size_t(ms.x) * ms.y * 16 * layers, nullptr));
            path_tracer_stage::options po = this->opt;
            po.distribution = dists[(size_t)rank];
            sl.ray_tracer = std::make_unique<Pipeline>(dev, scene_update, sl.color, po);
            if(n_slots > 1) { sl.ray_tracer->set_frame_slots(n_slots); }
            if(rank == 0)
            {
                sl.display = dev.alloc(size_t(size.x) * size.y * 16 * layers);
                sl.partials.assign((size_t)nranks, nullptr);
                for(int r = 1; r < nranks; ++r)
                {
                    const uvec2 rs = get_distribution_target_max_size(dists[(size_t)r]);
                    sl.partials[(size_t)r] = dev.alloc(size_t(rs.x) * rs.y * 16 * layers);
                }
            }
        }
        dev.sync();
        if(rank == 0) tonemap = std::make_unique<tonemap_stage>(dev, this->opt.tonemap);
    }

    ~basic_process_rt_renderer()
    {
        finish_all();
        for(slot_data& sl: slots)
        {
            sl.ray_tracer.reset();
            for(void* p: sl.partials) if(p) dev.free(p);
            if(sl.display) dev.free(sl.display);
            dev.free(sl.color);
            dev.destroy_stream(sl.stream);
        }
        if(ipc) trhip_ipc_destroy(ipc);
        trhip_comm_destroy(comm);
    }

    // The partial frames travel on the copy engines instead of through RCCL's kernels (trhip_ipc_*, include/trhip_comm.h): every rank
    // writes its strip into the display rank's IPC-mapped arena with hipMemcpyAsync on its slot's stream, tags order it with the stitch.
    // `allgather`: the caller's transport for the 256-byte set-up blobs (allgather_blobs_through_files, MPI ...): rank-major result.
    // Every rank calls this once, before the first frame.
    template<typename F>
    void use_copy_engine_exchange(F&& allgather)
    {
        if(nranks == 1 || ipc) return;
        check_comm(trhip_ipc_create(hip_device, nranks, rank, 0, size_t(size.x) * size.y * 16 * layers + 4096 * 16 * layers, (int)slots.size(), &ipc));
        std::vector<char> blob(TRHIP_IPC_EXPORT_BYTES);
        check_comm(trhip_ipc_export(ipc, blob.data()));
        const std::vector<char> all = allgather(blob);
        if(all.size() != blob.size() * (size_t)nranks) throw std::runtime_error("use_copy_engine_exchange: the all-gather returned the wrong size");
        check_comm(trhip_ipc_connect(ipc, all.data()));
    }

    // Every rank of a job has to shade with the same program: the reference compiles one pipeline per stage from the options and every
    // device gets that one (src/path_tracer_stage.cc:30-116).  Here a rank whose run-time compilation failed, or that was started under
    // TRHIP_SPECIALIZE=0 / another build of the library, would render its strips with other kernels - at the default arithmetic another
    // implementation inside Vulkan's accuracy, i.e. strips that differ from their neighbours' in the last bits.
    // `allgather`: as for use_copy_engine_exchange (blobs of 256 bytes).  Every rank calls this once, before the first frame; a mismatch throws
    // on every rank, naming the two programs.
    template<typename F>
    void check_same_program(F&& allgather)
    {
        if(nranks == 1) return;
        const trhip_program_info mine = slots[0].ray_tracer->program();
        std::vector<char> blob(256, 0);
        std::memcpy(blob.data(), &mine.identity, 8);
        std::memcpy(blob.data() + 8, mine.key, std::min(sizeof(mine.key), blob.size() - 9));
        const std::vector<char> all = allgather(blob);
        if(all.size() != blob.size() * (size_t)nranks) throw std::runtime_error("check_same_program: the all-gather returned the wrong size");
        for(int r = 0; r < nranks; ++r)
        {
            uint64_t id; std::memcpy(&id, all.data() + (size_t)r * blob.size(), 8);
            if(id != mine.identity)
                throw std::runtime_error("the ranks of this job would render with different shading programs: rank " + std::to_string(rank) + " {" + mine.key +
                                         "} but rank " + std::to_string(r) + " {" + std::string(all.data() + (size_t)r * blob.size() + 8) + "} (same libtrhip.so, kernel cache and TRHIP_* environment on every rank?)");
        }
    }

    void reset_accumulation(bool reset_sample_counter = false)
    {
        for(slot_data& sl: slots)
        {
            sl.ray_tracer->reset_accumulated_samples();
            if(reset_sample_counter) sl.ray_tracer->reset_sample_counter();
        }
        if(reset_sample_counter) frame_index = 0;
        accumulated_frames = 0;
    }

    // (behind every wait: a frame of the copy-engine exchange whose device-side wait for a peer gave up is an error here, before the
    // caller saves or shows it - the last frame of a job is followed by no other call into the exchange: trhip_ipc_check)
    void finish_frame() { if(current_slot >= 0) dev.sync(slots[(size_t)current_slot].stream); check_exchange(); }
    void finish_slot(int k) { dev.sync(slots[(size_t)k].stream); check_exchange(); }
    void finish_all() { for(slot_data& sl: slots) dev.sync(sl.stream); dev.sync(); }
    void check_exchange() { if(ipc) check_comm(trhip_ipc_check(ipc)); }

    // rt_renderer::render (src/rt_renderer.cc:84-133) for this rank.  Every rank calls it once per frame, in the same order.
    void render()
    {
        const size_t k = frame_index % slots.size();
        current_slot = (int)k;
        slot_data& sl = slots[k];
        if(!opt.accumulate) sl.ray_tracer->reset_accumulated_samples();
        if(slots.size() > 1) sl.ray_tracer->set_frame_counter(frame_index);
        sl.ray_tracer->run(sl.stream);
        if(nranks > 1)
        {
            // device_transfer: one grouped exchange on the slot's stream, behind the path tracing it ships and - on rank 0 - in
            // front of the stitch that reads it
            std::vector<size_t> bytes((size_t)nranks, 0);
            for(int r = 0; r < nranks; ++r) bytes[(size_t)r] = target_bytes(dists[(size_t)r]);
            std::vector<void*> arrived;      // copy-engine exchange: where the partial frames are in the arena
            if(ipc)
            {
                arrived.assign((size_t)nranks, nullptr);
                check_comm(trhip_ipc_gather_partials(ipc, sl.color, bytes[(size_t)rank], rank == 0 ? arrived.data() : nullptr, rank == 0 ? bytes.data() : nullptr, sl.stream));
            }
            else if(comm)
                check_comm(trhip_gather_partials(comm, 0, sl.color, bytes[(size_t)rank], rank == 0 ? sl.partials.data() : nullptr,
                                                 rank == 0 ? bytes.data() : nullptr, sl.stream));
            else throw std::runtime_error("process_rt_renderer: no exchange (a communicator id or use_copy_engine_exchange)");
            if(rank == 0)
            {
                std::vector<trhip_distribution> ds;
                std::vector<const void*> ps;
                std::vector<uint32_t> ws, hs;
                for(int r = 1; r < nranks; ++r)
                {
                    const uvec2 ts = get_distribution_target_size(dists[(size_t)r]);
                    if(ts.x == 0 || ts.y == 0) continue;
                    ds.push_back(to_abi(dists[(size_t)r])); ps.push_back(ipc ? arrived[(size_t)r] : sl.partials[(size_t)r]); ws.push_back(ts.x); hs.push_back(ts.y);
                }
                if(!ds.empty())
                    check(trhip_stitch_batch(dev.h, (uint32_t)ds.size(), ds.data(), ps.data(), ws.data(), hs.data(), sl.color, (uint32_t)layers,
                                             stitch_blend_ratio, sl.stream));
                stitch_blend_ratio = 1.0f;      // src/rt_renderer.cc:122
                if(ipc) check_comm(trhip_ipc_release(ipc, sl.stream));      // behind the stitch: the senders may overwrite the slot
            }
        }
        if(rank == 0)
        {
            display = sl.display;
            tonemap->run(sl.color, display, size, (uint32_t)layers, sl.stream);
        }
        frame_index++;
        accumulated_frames++;
    }

    // rt_renderer::set_device_workloads (src/rt_renderer.cc:135-183); every rank calls it with the same ratios
    void set_device_workloads(const std::vector<double>& ratios)
    {
        if(opt.distribution.strategy != DISTRIBUTION_SHUFFLED_STRIPS) return;
        finish_all();
        set_dists(ratios);
        for(slot_data& sl: slots)
        {
            sl.ray_tracer->reset_distribution_params(dists[(size_t)rank]);
            if(rank != 0) sl.ray_tracer->reset_accumulated_samples();
        }
        if(opt.accumulate && nranks > 1) stitch_blend_ratio = 1.0f / float(accumulated_frames + 1);
    }

    double get_path_tracing_time() { return slots[current_slot < 0 ? 0 : (size_t)current_slot].ray_tracer->get_duration_ms(); }

    struct slot_data
    {
        void* stream = nullptr;
        std::unique_ptr<Pipeline> ray_tracer;
        void* color = nullptr;
        void* display = nullptr;                 // rank 0
        std::vector<void*> partials;             // rank 0: where rank r's partial frame lands
    };
    int rank, nranks;
    uvec2 size;
    options opt;
    device dev;
    scene_stage scene_update;
    trhip_comm* comm = nullptr;
    trhip_ipc* ipc = nullptr;
    int hip_device = 0;
    std::vector<distribution_params> dists;      // every rank's share (all ranks compute all of them)
    std::vector<slot_data> slots;
    std::unique_ptr<tonemap_stage> tonemap;
    size_t layers = 1;
    int current_slot = -1;
    uint32_t frame_index = 0;
    unsigned accumulated_frames = 0;
    float stitch_blend_ratio = 1.0f;
    void* display = nullptr;

private:
    size_t target_bytes(const distribution_params& d) const { const uvec2 ts = get_distribution_target_size(d); return size_t(ts.x) * ts.y * 16 * layers; }
    void set_dists(const std::vector<double>& ratios)
    {
        if((int)ratios.size() != nranks) throw std::runtime_error("set_device_workloads needs one ratio per rank");
        dists.resize((size_t)nranks);
        double cumulative = 0;
        for(int r = 0; r < nranks; ++r)
        {
            const double ratio = std::min(std::max(ratios[(size_t)r], 0.0), 1.0 - cumulative);
            dists[(size_t)r] = get_device_distribution_params(size, opt.distribution.strategy, cumulative, ratio, (unsigned)r, (unsigned)nranks, r == 0);
            cumulative += ratio;
        }
    }
};
using process_rt_renderer = basic_process_rt_renderer<path_tracer_stage>;

}  // namespace tr
