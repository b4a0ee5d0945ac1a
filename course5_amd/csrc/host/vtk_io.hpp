// Built-in readers/writers for the two file formats of the `course` contract, so the binary
// does not need VTK (absent in this image):
//   in : legacy VTK "DATASET UNSTRUCTURED_GRID" (ASCII or BINARY, classic CELLS or the 5.1
//        OFFSETS/CONNECTIVITY layout), what vtkUnstructuredGridReader reads at
//        object3d_base.cpp:3-11;
//   out: VTK XML ImageData (.vti) with one 2-component Float64 point array "ImageScalars",
//        what object2d::export_to_vti writes (object2d.cpp:7-29).
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

struct vtk_grid {
    std::vector<double> points;                  // xyz per point
    std::vector<int32_t> tets;                   // first four point ids of every cell (object3d_base.cpp:37-42)
    std::map<std::string, std::vector<double>> cell_scalars;  // first component per cell (object3d_base.cpp:45-47)
    int64_t n_points() const { return static_cast<int64_t>(points.size() / 3); }
    int64_t n_cells() const { return static_cast<int64_t>(tets.size() / 4); }
};

// Throws std::runtime_error with a readable message on malformed input.
vtk_grid read_legacy_vtk(const std::string& path);

// image[row][col][2] fp32 (col fastest) -> .vti with dims (res_x, res_y, 1), origin 0, spacing 1,
// Float64 x 2 "ImageScalars" (object2d.cpp:12-13).
//   compressed = true : appended base64 data, vtkZLibDataCompressor blocks of 32 KiB — the encoding
//                       vtkXMLImageDataWriter uses by default, i.e. what the reference writes;
//   compressed = false: appended raw data (larger, fastest to write).
void write_vti(const std::string& path, const float* image, int res_x, int res_y, bool compressed = true);

// Colour-mapped 8-bit RGB PNG of one channel (0: tau, 1: I — the component utility/screen.py colours by),
// "Cool to Warm" between lo and hi, NaN pixels yellow (ParaView's defaults); scanlines from the top image row
// (largest y) down.  colour_range: smallest and largest finite value of the channel (ParaView's rescale to
// the data range on load).
void colour_range(const float* image, int res_x, int res_y, int channel, double* lo, double* hi);
void write_png(const std::string& path, const float* image, int res_x, int res_y, int channel, double lo, double hi);
