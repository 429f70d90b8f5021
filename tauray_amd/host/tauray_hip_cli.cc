// tauray_hip - headless command line front-end of the MI355X path-tracing core, shaped after `tauray --headless`
// (reference src/main.cc, src/tauray.cc:1017-1132 replay_viewer): load a scene, create an rt_renderer over the
// selected devices, render N frames, tonemap, save.  Scene input is a .glb / .gltf file (include/tauray_gltf.hh, the loader of
// src/gltf.cc for the path tracer's subset) or a .trsc dump (tauray_amd/scene_io.py); --dump-scene=out.trsc writes the loaded
// scene as a dump and exits without touching a GPU.
//
//   tauray_hip scene.glb|scene.trsc --width=512 --height=512 --headless=out/frame [--max-ray-depth=8] [--samples-per-pixel=1]
//              [--frames=1] [--fake-devices=N | --devices=0,1,...] [--distribution-strategy=scanline|shuffled-strips]
//              [--filetype=exr|raw|none] [--format=rgb16|rgb32|rgba16|rgba32] [--tonemap=filmic|linear|gamma-correction|
//              reinhard|reinhard-luminance] [--exposure=1] [--gamma=2.2] [--sampler=uniform-random|sobol-owen|sobol-z2|sobol-z3]
//              [--rng-seed=0] [--accumulation] [-t] [--skip-nan-check] [--warmup-frames=0] [--frames-in-flight=1] [--frames-per-launch=1]
//              [--renderer=path-tracer|direct]
//              [--camera-grid=w,h,x,y --camera-recentering-distance=5 --camera-grid-roll=0]   (light-field grid, one file per view)
//
// One process per GPU (include/tauray_hip_comm.hh): start N copies with --process-count=N --process-rank=0..N-1 --device=<HIP index>
// --comm-id=<file on a shared file system> [--comm-nonce=<number every rank of this job gets, e.g. the launcher's pid>] (rank 0 writes the RCCL id
// there under that nonce, the others wait for a file that carries it; rank 0 removes the file once the communicator exists); --exchange=ipc moves the
// partial frames on the copy engines instead of through RCCL (trhip_ipc_*: the set-up blobs travel through <file>.ipc<rank>); every rank renders its share
// of each frame, the partial frames meet on rank 0 through trhip_gather_partials, rank 0 stitches, tonemaps and saves.  With
// --shard=views the ranks divide the viewports of a camera grid instead (viewport v on rank v mod N) and save their own views.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>

#include "tauray_hip.hh"
#include "tauray_hip_comm.hh"
#include "tauray_envmap.hh"
#include "tauray_gltf.hh"

using namespace tr;

static bool starts(const std::string& s, const std::string& p) { return s.compare(0, p.size(), p) == 0; }

int main(int argc, char** argv)
{
    // hardware queues of the HIP runtime (default 4): more than three frame slots only pay off with more; read when the
    // runtime starts, i.e. before the first call into libtrhip
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    // memory shared between processes (RCCL, the copy-engine exchange): dmabuf IPC is what the target boxes' host driver supports
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    try
    {
        std::string scene_path, prefix = "capture";
        uvec2 size{1280, 720};
        int frames = 1, warmup = 0, fake_devices = 1, frames_in_flight = 1, frames_per_launch = 1;
        std::string renderer = "path-tracer";
        std::vector<int> devices;
        bool timing = false;
        std::string dump_scene;
        std::string envmap_path;
        bool frames_given = false, animation_flag = false;      // --animation[=name] --framerate=F (src/options.hh:110-129)
        std::string animation_name;
        double framerate = 60.0;
        int grid_w = 1, grid_h = 1;                                         // --camera-grid=w,h,x,y (src/options.hh camera_grid; generate_cameras)
        double grid_dx = 0, grid_dy = 0, grid_recentering = 5.0, grid_roll = 0;
        bool shard_views = false;                                           // --shard=views with --process-count: viewport v on rank v mod N
        int process_rank = -1, process_count = 0, process_device = 0;      // one process per GPU (see above)
        std::string comm_id_path;
        uint64_t comm_nonce = 0;
        std::string exchange = "rccl";
        std::vector<double> workloads;      // --device-workloads=a,b,...: rt_renderer::set_device_workloads before the first frame
        rt_renderer::options opt;
        opt.distribution.strategy = DISTRIBUTION_SHUFFLED_STRIPS;      // CLI default (src/options.hh:43-49)
        headless::options hopt;
        hopt.single_frame = true;
        for(int i = 1; i < argc; ++i)
        {
            std::string a = argv[i];
            auto val = [&](const char* key) { return a.substr(std::strlen(key)); };
            if(a == "-t") timing = true;
            else if(a == "--skip-nan-check") hopt.skip_nan_check = true;     // headless::options::skip_nan_check (src/headless.hh:74); with --filetype=none: no readback at all
            else if(a == "--accumulation") opt.accumulate = true;
            else if(a == "--pre-transform-vertices") opt.pre_transformed_vertices = true;
            else if(starts(a, "--width=")) size.x = (uint32_t)std::stoul(val("--width="));
            else if(starts(a, "--height=")) size.y = (uint32_t)std::stoul(val("--height="));
            else if(starts(a, "--headless=")) prefix = val("--headless=");
            else if(starts(a, "--max-ray-depth=")) opt.max_ray_depth = std::stoi(val("--max-ray-depth="));
            else if(starts(a, "--min-ray-dist=")) opt.min_ray_dist = std::stof(val("--min-ray-dist="));
            else if(starts(a, "--samples-per-pixel=")) opt.samples_per_pixel = std::stoi(val("--samples-per-pixel="));
            else if(starts(a, "--samples-per-pass=")) opt.samples_per_pass = std::stoi(val("--samples-per-pass="));
            else if(starts(a, "--frames=")) { frames = std::stoi(val("--frames=")); hopt.single_frame = frames == 1; frames_given = true; }
            else if(a == "--animation") animation_flag = true;                                  // any clip a node has (src/tauray.cc:252-253)
            else if(starts(a, "--animation=")) { animation_flag = true; animation_name = val("--animation="); }
            else if(starts(a, "--framerate=")) framerate = std::stod(val("--framerate="));
            else if(starts(a, "--envmap=")) envmap_path = val("--envmap=");                        // lat-long .hdr (src/options.hh:125)
            else if(starts(a, "--warmup-frames=")) warmup = std::stoi(val("--warmup-frames="));
            else if(starts(a, "--renderer="))
            {
                renderer = val("--renderer=");
                if(renderer != "path-tracer" && renderer != "direct") throw std::runtime_error("unknown renderer " + renderer + " (path-tracer, direct)");
            }
            else if(starts(a, "--dump-scene=")) dump_scene = val("--dump-scene=");
            else if(starts(a, "--device-workloads="))
            {
                std::string v = val("--device-workloads=");
                for(size_t p0 = 0; p0 <= v.size();)
                {
                    size_t p1 = v.find(',', p0);
                    if(p1 == std::string::npos) p1 = v.size();
                    workloads.push_back(std::stod(v.substr(p0, p1 - p0)));
                    p0 = p1 + 1;
                }
            }
            else if(starts(a, "--frames-in-flight=")) frames_in_flight = std::max(1, std::stoi(val("--frames-in-flight=")));
            else if(starts(a, "--frames-per-launch=")) frames_per_launch = std::max(1, std::stoi(val("--frames-per-launch=")));   // rt_renderer::options::frames_per_launch
            else if(starts(a, "--camera-grid="))
            {
                std::stringstream ss(val("--camera-grid=")); std::string tok; std::vector<double> v;
                while(std::getline(ss, tok, ',')) v.push_back(std::stod(tok));
                if(v.size() != 4 || v[0] < 1 || v[1] < 1) throw std::runtime_error("--camera-grid=w,h,x,y");
                grid_w = (int)v[0]; grid_h = (int)v[1]; grid_dx = v[2]; grid_dy = v[3];
            }
            else if(starts(a, "--camera-recentering-distance=")) grid_recentering = std::stod(val("--camera-recentering-distance="));
            else if(starts(a, "--camera-grid-roll=")) grid_roll = std::stod(val("--camera-grid-roll="));
            else if(starts(a, "--shard="))
            {
                const std::string v = val("--shard=");
                if(v != "views" && v != "pixels") throw std::runtime_error("--shard=pixels|views");
                shard_views = v == "views";
            }
            else if(starts(a, "--process-rank=")) process_rank = std::stoi(val("--process-rank="));
            else if(starts(a, "--process-count=")) process_count = std::stoi(val("--process-count="));
            else if(starts(a, "--device=")) process_device = std::stoi(val("--device="));
            else if(starts(a, "--comm-id=")) comm_id_path = val("--comm-id=");
            else if(starts(a, "--comm-nonce=")) comm_nonce = std::stoull(val("--comm-nonce="));
            else if(starts(a, "--exchange=")) exchange = val("--exchange=");
            else if(starts(a, "--fake-devices=")) fake_devices = std::stoi(val("--fake-devices="));
            else if(starts(a, "--rng-seed=")) opt.rng_seed = std::stoi(val("--rng-seed="));
            else if(starts(a, "--exposure=")) opt.tonemap.exposure = std::stof(val("--exposure="));
            else if(starts(a, "--gamma=")) opt.tonemap.gamma = std::stof(val("--gamma="));
            else if(starts(a, "--devices="))
            {
                std::stringstream ss(val("--devices=")); std::string tok;
                while(std::getline(ss, tok, ',')) devices.push_back(std::stoi(tok));
            }
            else if(starts(a, "--distribution-strategy="))
            {
                std::string v = val("--distribution-strategy=");
                opt.distribution.strategy = v == "scanline" ? DISTRIBUTION_SCANLINE : v == "duplicate" ? DISTRIBUTION_DUPLICATE : DISTRIBUTION_SHUFFLED_STRIPS;
            }
            else if(starts(a, "--filetype="))
            {
                std::string v = val("--filetype=");
                hopt.output_file_type = v == "raw" ? headless::RAW : (v == "none" ? headless::EMPTY : headless::EXR);
            }
            else if(starts(a, "--compression="))
            {
                static const std::map<std::string, tr::headless::compression_type> comps = {
                    {"none", tr::headless::NONE}, {"rle", tr::headless::RLE}, {"zips", tr::headless::ZIPS}, {"zip", tr::headless::ZIP}, {"piz", tr::headless::PIZ}};
                auto it = comps.find(val("--compression="));
                if(it == comps.end()) throw std::runtime_error("unknown compression " + val("--compression="));
                hopt.output_compression = it->second;
            }
            else if(starts(a, "--format="))
            {
                std::string v = val("--format=");
                hopt.output_format = v == "rgb32" ? headless::RGB32 : v == "rgba16" ? headless::RGBA16 : v == "rgba32" ? headless::RGBA32 : headless::RGB16;
            }
            else if(starts(a, "--tonemap="))
            {
                static const std::map<std::string, tonemap_stage::operator_type> ops = {
                    {"linear", tonemap_stage::LINEAR}, {"gamma-correction", tonemap_stage::GAMMA_CORRECTION}, {"filmic", tonemap_stage::FILMIC},
                    {"reinhard", tonemap_stage::REINHARD}, {"reinhard-luminance", tonemap_stage::REINHARD_LUMINANCE}};
                auto it = ops.find(val("--tonemap="));
                if(it == ops.end()) throw std::runtime_error("unknown tonemap operator");
                opt.tonemap.tonemap_operator = it->second;
            }
            else if(starts(a, "--sampler="))
            {
                std::string v = val("--sampler=");
                opt.local_sampler = v == "sobol-owen" ? sampler_type::SOBOL_OWEN : v == "sobol-z2" ? sampler_type::SOBOL_Z_ORDER_2D :
                    v == "sobol-z3" ? sampler_type::SOBOL_Z_ORDER_3D : sampler_type::UNIFORM_RANDOM;
            }
            else if(starts(a, "--")) throw std::runtime_error("unknown option " + a);
            else scene_path = a;
        }
        if(scene_path.empty()) throw std::runtime_error("usage: tauray_hip scene.glb|scene.trsc [options]");
        if(devices.empty()) devices.assign((size_t)std::max(fake_devices, 1), 0);

        const bool is_glb = (scene_path.size() > 4 && scene_path.compare(scene_path.size() - 4, 4, ".glb") == 0) ||
                            (scene_path.size() > 5 && scene_path.compare(scene_path.size() - 5, 5, ".gltf") == 0);
        scene_data scene = is_glb ? load_glb(scene_path, size.x, size.y) : load_scene_dump(scene_path);
        if(!envmap_path.empty()) set_envmap(scene, envmap_path);      // src/tauray.cc:198-201
        uint32_t viewports = 1;
        if(grid_w * grid_h > 1) viewports = generate_cameras(scene, grid_w, grid_h, grid_dx, grid_dy, grid_recentering, grid_roll);      // src/tauray.cc:680-727
        // play(scene, name, !replay, name == "") (src/tauray.cc:252-253); ticks in microseconds per update (:1052)
        scene_animator animator(scene);
        if(animation_flag) animator.play(animation_name, false);
        const int64_t update_dt = (int64_t)std::floor(1000000.0 / framerate + 0.5);
        const bool animated = animation_flag && animator.is_playing();
        if(animated && !frames_given) { frames = std::numeric_limits<int>::max(); hopt.single_frame = false; }      // until the clip ends
        if(!dump_scene.empty())
        {
            // with --animation the dump is the scene after --frames updates (the first one by dt = 0)
            if(animated) for(int f = 0; f < (frames_given ? frames : 1); ++f) animator.update(f == 0 ? 0 : update_dt);   // the flattened scene, and next to it what the loader found of skins: per skinned instance u32 instance, u32 vertices,
            // u32 joints, the {joints, weights} records, the rest-pose joint matrices
            write_scene_dump(scene, dump_scene);
            if(!scene.skinned.empty())
            {
                std::ofstream f(dump_scene + ".skins", std::ios::binary);
                for(const auto& sk: scene.skinned)
                {
                    const uint32_t head[3] = {sk.instance, (uint32_t)sk.skins.size(), (uint32_t)(sk.joint_transforms.size() / 16)};
                    f.write(reinterpret_cast<const char*>(head), 12);
                    f.write(reinterpret_cast<const char*>(sk.skins.data()), (std::streamsize)(sk.skins.size() * sizeof(trhip_skin)));
                    f.write(reinterpret_cast<const char*>(sk.joint_transforms.data()), (std::streamsize)(sk.joint_transforms.size() * 4));
                }
            }
            return 0;
        }
        // create_renderer (src/tauray.cc:355-421): classes without lights get weight 0, projection follows the camera
        if(scene.point_light_count() == 0) opt.sampling_weights.point_lights = 0;
        if(scene.directional_light_count() == 0) opt.sampling_weights.directional_lights = 0;
        if(scene.envmap.empty()) opt.sampling_weights.envmap = 0;
        if(!scene.has_tri_lights()) opt.sampling_weights.emissive_triangles = 0;
        opt.projection = (int)scene.projection;
        opt.samples_per_pass = std::min(opt.samples_per_pass, opt.samples_per_pixel);
        opt.active_viewport_count = viewports;

        opt.max_frames_in_flight = frames_in_flight;
        opt.frames_per_launch = frames_per_launch;
        hopt.size = size; hopt.output_prefix = prefix; hopt.display_count = viewports;
        if(shard_views)
        {   // view shards: viewport v belongs to rank v mod N; every rank renders, tonemaps and saves its own
            // views under their global indices, nothing is exchanged
            if(process_count < 1 || process_rank < 0 || process_rank >= process_count) throw std::runtime_error("--shard=views needs --process-count and --process-rank");
            if(animated || renderer != "path-tracer") throw std::runtime_error("--shard=views renders still frames with the path tracer");
            const uint32_t mine = viewports > (uint32_t)process_rank ? (viewports - (uint32_t)process_rank + (uint32_t)process_count - 1) / (uint32_t)process_count : 0;
            if(mine == 0) return 0;
            opt.active_viewport_count = mine;
            hopt.display_count = mine; hopt.display_count_total = viewports; hopt.display_index_base = (unsigned)process_rank; hopt.display_index_stride = (unsigned)process_count;
            headless vout(hopt);
            rt_renderer rr({process_device}, scene, size, opt);
            for(auto& sl: rr.per_device[0].slots) sl.ray_tracer->set_shard((uint32_t)process_rank, (uint32_t)process_count);
            for(int f = -warmup; f < frames; ++f)
            {
                rr.reset_accumulation();
                rr.render();
                rr.finish_frame();
                if(f >= 0) vout.save(*rr.per_device[0].dev, rr.display, (unsigned)f);
            }
            return 0;
        }
        headless out(hopt);
        if(process_count > 0)
        {   // one process per GPU: this process is rank process_rank of process_count
            if(process_rank < 0 || process_rank >= process_count) throw std::runtime_error("--process-rank must be in [0, --process-count)");
            if(comm_id_path.empty()) throw std::runtime_error("--process-count needs --comm-id=<file every rank can read>");
            if(renderer != "path-tracer" || animated) throw std::runtime_error("--process-count renders still frames with the path tracer");
            if(exchange != "rccl" && exchange != "ipc") throw std::runtime_error("--exchange is rccl or ipc");
            std::vector<char> id;
            if(exchange == "rccl") id = exchange_comm_id_through_file(comm_id_path, process_rank, comm_nonce);
            process_rt_renderer rr(process_device, process_rank, process_count, exchange == "rccl" ? id.data() : nullptr, scene, size, opt);
            if(exchange == "rccl") remove_comm_id_file(comm_id_path, process_rank);   // the communicator exists on every rank: the file has done its job
            else rr.use_copy_engine_exchange([&](const std::vector<char>& blob) { return allgather_blobs_through_files(comm_id_path, process_rank, process_count, blob, comm_nonce); });
            // all ranks shade with the same program, or none renders (needs the job's nonce like the blobs above; without one - a one-rank
            // job, or RCCL ranks started without --comm-nonce - there is nothing to tell this job's files from another's and the check is skipped with a note)
            if(process_count > 1 && comm_nonce != 0)
                rr.check_same_program([&](const std::vector<char>& blob) { return allgather_blobs_through_files(comm_id_path, process_rank, process_count, blob, comm_nonce, 120.0, ".prog"); });
            else if(process_count > 1 && process_rank == 0)
                std::cerr << "tauray_hip: no --comm-nonce: the ranks' shading programs are not compared (trhip_pt_get_program)\n";
            if(!workloads.empty()) rr.set_device_workloads(workloads);
            for(int f = -warmup; f < frames; ++f)
            {
                auto t0 = std::chrono::high_resolution_clock::now();
                rr.reset_accumulation();
                rr.render();
                rr.finish_frame();
                auto t1 = std::chrono::high_resolution_clock::now();
                if(f < 0) continue;
                if(timing)
                    std::cout << "FRAME " << f << ":\n\tRANK " << process_rank << ":\n\t\t[path tracing (" << opt.active_viewport_count << " viewports)] "
                              << rr.get_path_tracing_time() << " ms\n\tHOST: " << std::chrono::duration<double, std::milli>(t1 - t0).count() << " ms\n";
                if(process_rank == 0) out.save(rr.dev, rr.display, (unsigned)f);
            }
            return 0;
        }
        // --renderer picks the pipeline rt_renderer<Pipeline> is instantiated with (src/tauray.cc:355-421: path-tracer, direct)
        auto run = [&](auto& rr) -> int
        {
        if(!workloads.empty())
        {
            if(workloads.size() != rr.per_device.size()) throw std::runtime_error("--device-workloads needs one ratio per device");
            rr.set_device_workloads(workloads);
        }
        if((frames_in_flight > 1 || frames_per_launch > 1) && !animated)
        {   // frame f renders while the frames before it are read back, compressed and written (the reference overlaps
            // its save workers with the next frames the same way, src/headless.cc:349-422); with --frames-per-launch=B a slot
            // holds B consecutive frames, frame-major in its display image
            const int B = frames_per_launch, none = std::numeric_limits<int>::min();
            const size_t frame_bytes = size_t(size.x) * size.y * 16 * opt.active_viewport_count;
            std::vector<int> in_slot(frames_in_flight, none);      // the first frame of the launch a slot holds (negative: warm-up)
            auto retire = [&](int k) {
                if(in_slot[k] == none) return;
                rr.finish_slot(k);
                for(int b = 0; b < B; ++b)
                {
                    const int f = in_slot[k] + b;
                    if(f >= 0 && f < frames)
                        out.save(*rr.per_device[0].dev, static_cast<const char*>(rr.frame_slots[k].display) + size_t(b) * frame_bytes, (unsigned)f);
                }
                in_slot[k] = none;
            };
            auto t0 = std::chrono::high_resolution_clock::now();
            bool started = false;
            for(int f = -warmup; f < frames; f += B)
            {
                const int k = (int)((rr.frame_index / (uint32_t)B) % (uint32_t)frames_in_flight);
                retire(k);                               // the slot's previous frames must be on disk before it is reused
                if(!started && f >= 0) { rr.finish_all(); t0 = std::chrono::high_resolution_clock::now(); started = true; }   // -t: the warm-up is over
                rr.render();
                in_slot[k] = f;
            }
            for(int n = 0; n < frames_in_flight; ++n) retire((int)((rr.frame_index / (uint32_t)B + n) % (uint32_t)frames_in_flight));
            if(timing)
            {
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
                const int counted = ((frames + B - 1) / B) * B;      // whole launches, like the loop above
                std::cout << "FRAMES " << counted << " (" << frames_in_flight << " in flight, " << B << " per launch): " << ms << " ms, " << ms / counted << " ms per frame\n";
            }
            return 0;
        }
        for(int f = -warmup; f < frames; ++f)
        {
            if(animated && f >= 0)
            {   // update(s, dt, true) before the frame, the first one by dt = 0; without --frames the run ends with the clip
                // (src/tauray.cc:1064-1092)
                if(!frames_given && !animator.is_playing()) break;
                animator.update(f == 0 ? 0 : update_dt);
                if(!frames_given && !animator.is_playing()) break;
                rr.update_scene(scene, f % 3 == 2);      // an acceleration-structure update, every third frame a fast rebuild
            }
            auto t0 = std::chrono::high_resolution_clock::now();
            rr.reset_accumulation();                   // offline frames: accumulation reset, sample counter kept (src/tauray.cc:1101)
            rr.render();
            rr.finish_frame();
            auto t1 = std::chrono::high_resolution_clock::now();
            if(f < 0) continue;
            if(timing)
            {   // print_simple_trace (src/tracing.cc:247-279)
                std::cout << "FRAME " << f << ":\n";
                std::vector<double> pt = rr.get_path_tracing_times();
                for(size_t i = 0; i < pt.size(); ++i)
                    std::cout << "\tDEVICE " << i << ":\n\t\t[path tracing (" << opt.active_viewport_count << " viewports)] " << pt[i] << " ms\n";
                std::cout << "\tHOST: " << std::chrono::duration<double, std::milli>(t1 - t0).count() << " ms\n";
            }
            out.save(*rr.per_device[0].dev, rr.display, (unsigned)f);
        }
        return 0;
        };
        if(renderer == "direct")
        {
            direct_renderer::options dopt;
            static_cast<path_tracer_stage::options&>(dopt) = opt;
            dopt.tonemap = opt.tonemap; dopt.accumulate = opt.accumulate; dopt.max_frames_in_flight = opt.max_frames_in_flight; dopt.frames_per_launch = opt.frames_per_launch;
            direct_renderer rr(devices, scene, size, dopt);
            return run(rr);
        }
        rt_renderer rr(devices, scene, size, opt);
        return run(rr);
    }
    catch(std::exception& e)
    {
        std::cerr << e.what() << std::endl;
        return 1;
    }
}
