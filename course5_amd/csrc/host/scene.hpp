// Host-side scene objects behind the reference's interface names (object3d_base.hpp:21-41,
// object3d_*.hpp, tetra.hpp:8-10), re-designed for a GPU-resident renderer:
//   - a volume object keeps points + connectivity (the reference copies four points per cell and
//     forgets the ids, object3d_base.cpp:37-51; the walk needs them for face adjacency);
//   - a solid object is a tet soup produced by init_polar;
//   - rotate_around_* does not touch vertices on the host: it appends to the object's rotation
//     list, which the device applies every frame with the reference's arithmetic
//     (tetra.cpp:44-62 -> exact_kernels.hip).
#pragma once

#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "config.hpp"
#include "course5_hip.h"

enum class tetra_value : std::size_t { alpha = 0, solid_color = 0, Q = 1 };  // tetra.hpp:8
enum class tetra_type { transparent = 0, solid = 1 };                        // tetra.hpp:10

using point = std::array<double, 3>;

struct object3d_data {
    tetra_type kind = tetra_type::transparent;
    // transparent: indexed grid
    std::vector<double> points;   // xyz
    std::vector<int32_t> cells;   // 4 ids per cell
    std::vector<double> value0;   // alpha per cell
    std::vector<double> value1;   // Q per cell
    // solid: soup, 12 doubles per tet, one colour
    std::vector<double> soup;
    double colour = 0;
    std::vector<c5_rotation> rotations;  // applied in order, every frame, on the device

    std::size_t size() const { return kind == tetra_type::solid ? soup.size() / 12 : cells.size() / 4; }
};

class object3d_base {
public:
    object3d_base() = default;
    virtual ~object3d_base() = default;

    // object3d_base.cpp:13-53; scalar_labels = {alpha name, Q name}
    void read_vtk_file(const std::string& filename, const std::vector<std::string>& scalar_labels);

    // object3d_base.cpp:83-196: star-shaped solid around (x0,y0,z0) bounded by potential == level_value
    void init_polar(const std::function<double(std::array<double, 3>)>& potential_function, double x0, double y0,
                    double z0, double level_value, double step, double angle_step,
                    tetra_type arg_tetra_type = tetra_type::transparent, double tetra_v1 = 0, double tetra_v2 = 0);

    std::shared_ptr<object3d_data> get_pointer() { return _data; }

    virtual void rotate_around_x_axis(double angle);             // object3d_base.cpp:202-210
    virtual void rotate_around_y_axis(double angle, double x0);  // object3d_base.cpp:212-219
    void clear_rotations() { _data->rotations.clear(); }

protected:
    std::shared_ptr<object3d_data> _data = std::make_shared<object3d_data>();
};

class object3d_accretion_disk : public object3d_base {
public:
    object3d_accretion_disk() = default;
    explicit object3d_accretion_disk(const std::string& filename);  // object3d_accretion_disk.cpp:3-5
};

class object3d_roche_lobe : public object3d_base {
public:
    // object3d_roche_lobe.cpp:20-49
    object3d_roche_lobe(const point& pos_accretor, double dist, double donor_angle_around_y, double m_accretor,
                        double m_donor, double def_omega);
};

class object3d_sphere : public object3d_base {
public:
    object3d_sphere(const point& center, double R);  // object3d_sphere.cpp:11-17
};
