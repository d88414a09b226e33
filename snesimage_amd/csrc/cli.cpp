// snesimage_amd/csrc/cli.cpp — headless driver over the C ABI, keeping the reference's command line
// (src/config.rs:3-31: source_filename target_filename [-c N] [-s N] [-d] [--perceptual-palettes]
// [--nes]) and its JSON output (src/lib.rs:579-625, 1002).  The reference drives the optimizer from
// an SDL2 window (src/lib.rs:855-1038, out of scope); here the three phases advance on their own:
// TileAssignment (initialize_tiles) -> Clustering (recalculate_palettes) -> Optimization for
// --calls optimizer calls in the reference's slot order (src/lib.rs:881-933) -> write the JSON.
// Extra flags exist only because the reference is interactive and unseeded: --seed, --calls,
// --candidates, --device, --tile-palettes FILE (1024 bytes, replaces the mouse clicks of
// src/lib.rs:1005-1017), --resume FILE.json (start from a previous output instead of the k-means initialisers: the
// reference's TODO.md:38-39), --preview FILE.png (source | result side by side, the picture the SDL window of
// src/lib.rs:937-960 shows).  The source image is a PNG (png_io.hpp restates `image::open(..).into_rgba8()`,
// src/lib.rs:836, for that format), a raw RGBA8 file of 256*H*4 bytes, or `synth:SEED`.
// The host language the north star asks for is Rust; no Rust toolchain exists in this image, so the
// driver is C++ over the same extern "C" surface a Rust crate would bind.
#include "../../include/snesimage_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "png_io.hpp"

namespace {

void log_info(const std::string &msg) { // src/util.rs:9-22: [time][level][target] message
    char ts[64];
    time_t now = time(nullptr);
    strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", localtime(&now));
    printf("[%s][INFO][snesimage] %s\n", ts, msg.c_str());
    fflush(stdout);
}
[[noreturn]] void die(const std::string &msg) { // src/main.rs:16-19
    char ts[64];
    time_t now = time(nullptr);
    strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", localtime(&now));
    printf("[%s][ERROR][snesimage] Error running application: %s\n", ts, msg.c_str());
    exit(1);
}
std::string fmt_f64(double v) { // Rust's `{}` for f64: shortest representation that round-trips
    char buf[64];
    for (int prec = 1; prec <= 17; prec++) {
        snprintf(buf, sizeof buf, "%.*g", prec, v);
        if (strtod(buf, nullptr) == v) break;
    }
    return buf;
}
void usage() {
    fprintf(stderr,
            "Usage: snesimage_cli [OPTIONS] <SOURCE_FILENAME> <TARGET_FILENAME>\n\n"
            "Arguments:\n  <SOURCE_FILENAME>  PNG image, raw RGBA8 file (256 x H x 4 bytes) or synth:SEED\n  <TARGET_FILENAME>  JSON output\n\n"
            "Options:\n  -c, --subpalette-count <N>  [default: 1]\n  -s, --subpalette-size <N>   [default: 7]\n"
            "  -d, --dither\n      --perceptual-palettes\n      --nes\n"
            "      --calls <N>          optimizer calls to run [default: 0]\n      --candidates <N>     random candidates per call [default: 64]\n"
            "      --window <N>         optimizer calls scored per launch set (0 = adaptive, 1 = call by call; same result) [default: 0]\n"
            "      --seed <N>           candidate RNG seed [default: 1]\n      --device <N>         HIP device [default: 0]\n"
            "      --devices <A,B,..>   shard every call's candidates over these devices (RCCL inside the library)\n"
            "      --tile-palettes <F>  1024-byte tile->subpalette override\n      --resume <F>         start from the palette and tile palettes of a previous JSON output\n"
            "      --preview <F>        write source | result as a PNG\n"
            "      --reassign-tiles <K> every K sweeps of the palette, move each tile to the subpalette that reproduces it best\n"
            "                           (off by default; not in the reference: its TODO.md lists it as missing)\n"
            "      --decode-only        write the decoded source as raw RGBA8 to <TARGET_FILENAME> and stop (no GPU)\n  -h, --help\n  -V, --version\n");
}
// the flat integer array stored under `"key":[...]` in one of this driver's (or the reference's) JSON outputs
bool json_int_array(const std::string &path, const char *key, std::vector<long> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string text; char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    const std::string pat = std::string("\"") + key + "\"";
    size_t at = text.find(pat);
    if (at == std::string::npos) return false;
    at = text.find('[', at + pat.size());
    if (at == std::string::npos) return false;
    out.clear();
    for (size_t i = at + 1; i < text.size(); ) {
        const char ch = text[i];
        if (ch == ']') return true;
        if (ch == '-' || (ch >= '0' && ch <= '9')) { char *end = nullptr; out.push_back(strtol(text.c_str() + i, &end, 10)); i = (size_t)(end - text.c_str()); }
        else if (ch == ',' || ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t') i++;
        else return false; // nested arrays or anything else: not a flat integer array
    }
    return false;
}

void synth(uint64_t seed, uint32_t w, uint32_t h, std::vector<uint8_t> &out) {
    out.resize((size_t)w * h * 4);
    uint64_t s = seed;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint64_t z = (s += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            uint8_t *o = &out[4 * ((size_t)y * w + x)];
            o[0] = (uint8_t)(x + (z & 63)); o[1] = (uint8_t)(y + ((z >> 8) & 63)); o[2] = (uint8_t)((x + y) / 2 + ((z >> 16) & 63)); o[3] = 255;
        }
}

} // namespace

int main(int argc, char **argv) {
    std::vector<std::string> pos;
    uint32_t count = 1, size = 7, flags = 0, calls = 0, ncand = 64, reassign_every = 0, window = 0; // src/config.rs:13-18 defaults
    uint64_t seed = 1;
    int device = 0;
    std::string tile_file, preview_file, resume_file;
    std::vector<int> devices; // --devices: candidate sharding over several GPUs from this one process
    bool decode_only = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *name) -> const char * { if (i + 1 >= argc) { fprintf(stderr, "error: a value is required for '%s'\n", name); exit(2); } return argv[++i]; };
        if (a == "-c" || a == "--subpalette-count") count = (uint32_t)strtoul(need("--subpalette-count"), nullptr, 10);
        else if (a == "-s" || a == "--subpalette-size") size = (uint32_t)strtoul(need("--subpalette-size"), nullptr, 10);
        else if (a == "-d" || a == "--dither") flags |= SNES_DITHER;
        else if (a == "--perceptual-palettes") flags |= SNES_PERCEPTUAL;
        else if (a == "--nes") flags |= SNES_NES;
        else if (a == "--calls") calls = (uint32_t)strtoul(need("--calls"), nullptr, 10);
        else if (a == "--candidates") ncand = (uint32_t)strtoul(need("--candidates"), nullptr, 10);
        else if (a == "--seed") seed = strtoull(need("--seed"), nullptr, 0);
        else if (a == "--window") window = (uint32_t)strtoul(need("--window"), nullptr, 10);
        else if (a == "--device") device = atoi(need("--device"));
        else if (a == "--devices") { for (const char *q = need("--devices"); *q;) { char *end = nullptr; devices.push_back((int)strtol(q, &end, 10)); if (end == q) { fprintf(stderr, "error: invalid value for '--devices'\n"); return 2; } q = *end == ',' ? end + 1 : end; } }
        else if (a == "--tile-palettes") tile_file = need("--tile-palettes");
        else if (a == "--preview") preview_file = need("--preview");
        else if (a == "--reassign-tiles") reassign_every = (uint32_t)strtoul(need("--reassign-tiles"), nullptr, 10);
        else if (a == "--resume") resume_file = need("--resume");
        else if (a == "--decode-only") decode_only = true;
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (a == "-V" || a == "--version") { printf("snesimage 0.1.1 (%s)\n", snesimage_version()); return 0; }
        else if (!a.empty() && a[0] == '-' && a != "-") { fprintf(stderr, "error: unexpected argument '%s' found\n", a.c_str()); usage(); return 2; }
        else pos.push_back(a);
    }
    if (pos.size() != 2) { fprintf(stderr, "error: the following required arguments were not provided: <SOURCE_FILENAME> <TARGET_FILENAME>\n"); usage(); return 2; }
    const std::string source = pos[0], target = pos[1];
    // the JSON keeps 15 colours per subpalette (src/lib.rs:583-593): a larger subpalette cannot be read back from it
    if (!resume_file.empty() && size > 15) die("--resume needs --subpalette-size <= 15: the output keeps 15 colours per subpalette");

    log_info("Using source image: " + source); // src/lib.rs:834
    std::vector<uint8_t> rgba;
    uint32_t w = 256, h = 256;
    if (source.rfind("synth:", 0) == 0) synth(strtoull(source.c_str() + 6, nullptr, 0), w, h, rgba);
    else {
        FILE *f = fopen(source.c_str(), "rb");
        if (!f) die("No such file or directory: " + source);
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> file(n > 0 ? (size_t)n : 0);
        if (n <= 0 || fread(file.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); die("short read on " + source); }
        fclose(f);
        if (snes_png::is_png(file.data(), file.size())) {
            std::string err;
            if (!snes_png::decode(file.data(), file.size(), w, h, rgba, err)) die("Format error decoding Png: " + err);
            // src/lib.rs:838-840 tests `width != 256 && height != 256` and then indexes tiles with a hard-coded 32;
            // this build admits what that arithmetic handles: 256 wide, and the height the library accepts
            if (w != 256) die("Image size must be 256x256");
        } else {
            if (n % (256 * 4) != 0) die("Image size must be 256x256"); // src/lib.rs:838-840
            h = (uint32_t)(n / (256 * 4));
            rgba.swap(file);
        }
    }
    if (decode_only) {
        FILE *f = fopen(target.c_str(), "wb");
        if (!f) die("cannot create " + target);
        fwrite(rgba.data(), 1, rgba.size(), f);
        fclose(f);
        printf("%u %u\n", w, h);
        return 0;
    }
    if (!devices.empty()) device = devices[0];
    snesimage_ctx *ctx = nullptr;
    if (snesimage_create(rgba.data(), w, h, count, size, flags, device, &ctx) != 0) die(snesimage_last_error());
    if (!resume_file.empty()) { // palette + tile_palettes of an earlier output (src/lib.rs:579-625); the tiles follow from optimize()
        std::vector<long> pal, tp;
        if (!json_int_array(resume_file, "palette", pal) || !json_int_array(resume_file, "tile_palettes", tp)) die("cannot read palette and tile_palettes from " + resume_file);
        if (pal.size() != 16 * (size_t)count || tp.size() != 1024) die("resume file does not match --subpalette-count (palette must hold 16 entries per subpalette, tile_palettes 1024)");
        std::vector<uint8_t> tp8(1024), rgb5(3 * (size_t)count * size);
        for (size_t i = 0; i < 1024; i++) { if (tp[i] < 0 || tp[i] >= (long)count) die("resume file: tile palette out of range"); tp8[i] = (uint8_t)tp[i]; }
        for (uint32_t p = 0; p < count; p++)
            for (uint32_t i = 0; i < size; i++) { // slot 0 of every 16 is the transparent colour (src/lib.rs:583-585)
                const long v = pal[16 * (size_t)p + 1 + i];
                if (v < 0 || v > 0x7fff) die("resume file: colour out of range");
                uint8_t *o = &rgb5[3 * ((size_t)p * size + i)];
                o[0] = (uint8_t)(v & 31); o[1] = (uint8_t)((v >> 5) & 31); o[2] = (uint8_t)((v >> 10) & 31);
            }
        if (snesimage_set_tile_palettes(ctx, tp8.data()) != 0 || snesimage_set_palette_rgb5(ctx, rgb5.data()) != 0 || snesimage_optimize(ctx) != 0) die(snesimage_last_error());
        log_info("Resumed from " + resume_file);
    } else {
        if (snesimage_initialize_tiles(ctx) != 0) die(std::string("Unable to initialize tiles: ") + snesimage_last_error()); // src/lib.rs:851-853
        log_info("Finished assigning initial tiles");
        if (!tile_file.empty()) {
            std::vector<uint8_t> tp(1024);
            FILE *f = fopen(tile_file.c_str(), "rb");
            if (!f || fread(tp.data(), 1, 1024, f) != 1024) die("cannot read 1024 bytes from " + tile_file);
            fclose(f);
            if (snesimage_set_tile_palettes(ctx, tp.data()) != 0) die(snesimage_last_error());
        }
        log_info("Generating initial palettes"); // src/lib.rs:985-989
        if (snesimage_recalculate_palettes(ctx) != 0) die(std::string("Unable to recalculate palettes: ") + snesimage_last_error());
    }
    // --devices: replicas of the initialised state on the other devices, and the group that shards each call over them
    std::vector<snesimage_ctx *> members{ctx};
    snesimage_group *group = nullptr;
    if (!devices.empty()) {
        std::vector<uint8_t> tp(1024), pal(3 * (size_t)count * size);
        if (snesimage_get_tile_palettes(ctx, tp.data()) != 0 || snesimage_get_palette_rgb5(ctx, pal.data()) != 0) die(snesimage_last_error());
        for (size_t d = 1; d < devices.size(); d++) {
            snesimage_ctx *m = nullptr;
            if (snesimage_create(rgba.data(), w, h, count, size, flags, devices[d], &m) != 0 || snesimage_set_tile_palettes(m, tp.data()) != 0 || snesimage_set_palette_rgb5(m, pal.data()) != 0 ||
                snesimage_optimize(m) != 0)
                die(snesimage_last_error());
            members.push_back(m);
        }
        if (snesimage_group_create(members.data(), (uint32_t)members.size(), &group) != 0) die(snesimage_last_error());
        log_info("Sharding candidates over " + std::to_string(members.size()) + " device(s)"); // (with the reference's 64 candidates per call: the calls of every window)
    }
    log_info("Beginning optimization"); // src/lib.rs:992
    uint32_t palette = 0, index = 0, channel = 0, step = 0, sweep = 0;
    double last_error = 1.7976931348623157e308;
    std::vector<uint8_t> before(3 * (size_t)count * size), after(before.size());
    const int nes = (flags & SNES_NES) ? 1 : 0;
    auto report = [&](uint32_t p, uint32_t ix, const uint8_t *b, const uint8_t *best, double error) {
        if (b[0] != best[0] || b[1] != best[1] || b[2] != best[2]) { // src/lib.rs:222-234
            char m[160];
            snprintf(m, sizeof m, "Setting color (%u, %u) from (%u, %u, %u) to (%u, %u, %u)", p, ix, b[0], b[1], b[2], best[0], best[1], best[2]);
            log_info(m);
        }
        if (std::abs(error - last_error) > 2.220446049250313e-16) { log_info("Current Error: " + fmt_f64(error)); last_error = error; } // src/lib.rs:912-915
    };
    auto end_of_sweep = [&]() {
        if (reassign_every && step != sweep && step % reassign_every == 0) { // a sweep over every slot has just ended (src/lib.rs:925-931)
            uint32_t moved = 0;
            for (snesimage_ctx *m : members) if (snesimage_reassign_tiles(m, &moved) != 0) die(std::string("Unable to reassign tiles: ") + snesimage_last_error());
            log_info("Reassigned " + std::to_string(moved) + " tiles");
        }
        sweep = step;
    };
    if (ncand <= 64 && window != 1) {
        // The reference's loop (src/lib.rs:888-933), several calls per launch: snesimage_run_slots scores the coming calls of the
        // schedule against the current palette and applies them in order up to the first one that changes it — the same
        // trajectory, call for call, as stepping one call at a time (--window 1).  A run ends with its sweep when tiles are
        // to be reassigned between sweeps.
        std::vector<snesimage_call_result> log;
        if (snesimage_get_palette_rgb5(ctx, before.data()) != 0) die(snesimage_last_error());
        snesimage_run_stats total{};
        for (uint32_t call = 0; call < calls;) {
            uint32_t n = calls - call;
            if (n > 4096) n = 4096;
            if (reassign_every) { // calls left in the current sweep
                uint32_t p = palette, ix = index, ch = channel, st = step, m = 0, k = 0;
                while (st == step && k < n) { snesimage_schedule_next(count, size, nes, &p, &ix, &ch, &st, &m); k++; }
                n = k;
            }
            log.resize(n);
            uint32_t p = palette, ix = index, ch = channel, st = step, method = 0;
            snesimage_run_stats rs{};
            if ((group ? snesimage_group_run_slots(group, n, seed, call, &palette, &index, &channel, &step, ncand, window, log.data(), &rs)
                       : snesimage_run_slots(ctx, n, seed, call, &palette, &index, &channel, &step, ncand, window, log.data(), &rs)) != 0) die(std::string("Unable to optimize palette: ") + snesimage_last_error());
            total.calls += rs.calls; total.accepted += rs.accepted; total.windows += rs.windows; total.scored += rs.scored; total.useful += rs.useful;
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t cp = p, ci = ix;
                snesimage_schedule_next(count, size, nes, &p, &ix, &ch, &st, &method);
                uint8_t *b = &before[3 * ((size_t)cp * size + ci)];
                report(cp, ci, b, log[j].rgb5, log[j].error);
                b[0] = log[j].rgb5[0]; b[1] = log[j].rgb5[1]; b[2] = log[j].rgb5[2];
            }
            call += n;
            end_of_sweep();
        }
        if (calls) {
            char m[200];
            snprintf(m, sizeof m, "Ran %u calls in %u launch sets: %u changed the palette, %llu of %llu candidates scored were of calls that took effect", total.calls, total.windows, total.accepted,
                     (unsigned long long)total.useful, (unsigned long long)total.scored);
            log_info(m);
        }
    } else
    for (uint32_t call = 0; call < calls; call++) {
        uint32_t p = palette, ix = index, ch = channel, method = 0;
        snesimage_schedule_next(count, size, nes, &palette, &index, &channel, &step, &method);
        snesimage_get_palette_rgb5(ctx, before.data());
        double error = 0.0; uint8_t best[3];
        const int32_t rc = group ? snesimage_group_step(group, method, p, ix, ch, seed, call, method == SNES_METHOD_RANDOM ? ncand : 0, &error, best)
                                 : snesimage_step(ctx, method, p, ix, ch, seed, call, method == SNES_METHOD_RANDOM ? ncand : 0, &error, best);
        if (rc != 0) die(std::string("Unable to optimize palette: ") + snesimage_last_error());
        report(p, ix, &before[3 * ((size_t)p * size + ix)], best, error);
        end_of_sweep();
    }
    log_info("Writing output to " + target); // src/lib.rs:1000-1002
    int64_t need = snesimage_as_json(ctx, nullptr, 0);
    if (need < 0) die(snesimage_last_error());
    std::vector<char> json((size_t)need);
    snesimage_as_json(ctx, json.data(), need);
    FILE *f = fopen(target.c_str(), "wb");
    if (!f) die("cannot create " + target);
    fwrite(json.data(), 1, (size_t)need - 1, f);
    fclose(f);
    if (!preview_file.empty()) { // left: source, right: as_rgba() of the result (src/lib.rs:940-957)
        std::vector<uint8_t> result((size_t)w * h * 4), both((size_t)2 * w * h * 4), png;
        if (snesimage_as_rgba(ctx, result.data()) != 0) die(snesimage_last_error());
        for (uint32_t y = 0; y < h; y++) {
            memcpy(&both[(size_t)y * 2 * w * 4], &rgba[(size_t)y * w * 4], (size_t)w * 4);
            memcpy(&both[((size_t)y * 2 + 1) * w * 4], &result[(size_t)y * w * 4], (size_t)w * 4);
        }
        if (!snes_png::encode_rgba(2 * w, h, both.data(), png)) die("cannot encode " + preview_file);
        FILE *pf = fopen(preview_file.c_str(), "wb");
        if (!pf) die("cannot create " + preview_file);
        fwrite(png.data(), 1, png.size(), pf);
        fclose(pf);
        log_info("Wrote preview to " + preview_file);
    }
    if (group) snesimage_group_destroy(group);
    for (snesimage_ctx *m : members) snesimage_destroy(m);
    return 0;
}
