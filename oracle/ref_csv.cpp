// oracle/ref_csv.cpp — TEST INFRASTRUCTURE, not part of the product.
//
// Builds the reference's OWN CSV writer: Loader/CSVWriter.h is plain C++ (std only) and compiles from where it lies
// under /root/reference (path passed as -DREF_CSV_HEADER=...; nothing is copied; output only into oracle/_ref/).
// The rows are assembled the way the reference's call sites assemble them — kernel.cu:61 (header), kernel.cu:18,37
// (build row {"", "", "", to_string(build)}), core/render.h:217-230 (frame rows {to_string(frame), to_string(render),
// to_string(update), ""}; a build row precedes the frame rows: renderBVH times the build before the animation loop) —
// and written by the reference's writeCSV.  tests/test_png.py compares crt_write_timing_csv with it byte for byte.
#include REF_CSV_HEADER

extern "C" int ref_write_csv(const char* path, int n_frames, const int* frame, const double* rendering, const double* update,
                             double build) {
    std::vector<std::vector<std::string>> data;
    data.push_back({"frame", "rendering", "update", "build"});
    data.push_back({"", "", "", std::to_string(build)});
    for (int i = 0; i < n_frames; ++i)
        data.push_back({std::to_string(frame[i]), std::to_string(rendering[i]), std::to_string(update[i]), ""});
    writeCSV(path, data);
    return 0;
}
