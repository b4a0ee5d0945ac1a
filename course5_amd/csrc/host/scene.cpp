#include "scene.hpp"

#include <cmath>
#include <limits>
#include <stdexcept>

#include "vtk_io.hpp"

// ---------------------------------------------------------------------------------------------
// object3d_base
// ---------------------------------------------------------------------------------------------
void object3d_base::read_vtk_file(const std::string& filename, const std::vector<std::string>& scalar_labels) {
    vtk_grid g = read_legacy_vtk(filename);
    object3d_data& d = *_data;
    d.kind = tetra_type::transparent;
    d.points = std::move(g.points);
    d.cells = std::move(g.tets);
    std::vector<double>* dst[2] = {&d.value0, &d.value1};
    for (std::size_t k = 0; k < 2; ++k) {
        dst[k]->assign(d.cells.size() / 4, 0.0);
        if (k >= scalar_labels.size()) continue;
        auto it = g.cell_scalars.find(scalar_labels[k]);
        // the reference dereferences a null vtkDataArray here (object3d_base.cpp:24-26,45-47); say why instead
        if (it == g.cell_scalars.end())
            throw std::runtime_error("cell scalar array '" + scalar_labels[k] + "' not found in " + filename);
        *dst[k] = it->second;
    }
}

void object3d_base::rotate_around_x_axis(double angle) {
    c5_rotation r{};
    r.axis = 0;
    r.angle = angle;
    r.x0 = 0;
    _data->rotations.push_back(r);
}

void object3d_base::rotate_around_y_axis(double angle, double x0) {
    c5_rotation r{};
    r.axis = 1;
    r.angle = angle;
    r.x0 = x0;
    _data->rotations.push_back(r);
}

namespace {

// object3d_base.cpp:55-65
point turn_about_y(const point& v, double a) {
    return point{v[0] * cos(a) + v[2] * sin(a), v[1], -v[0] * sin(a) + v[2] * cos(a)};
}
// object3d_base.cpp:67-75
point turn_about_z(const point& v, double a) {
    return point{v[0] * cos(a) + v[1] * sin(a), -v[0] * sin(a) + v[1] * cos(a), v[2]};
}
void advance(point& p, const point& d) {
    for (std::size_t k = 0; k < 3; ++k) p[k] += d[k];
}

// march from `origin` along `dir` until the potential reaches the level (object3d_base.cpp:101-107,133-136)
point march_to_level(const std::function<double(std::array<double, 3>)>& f, const point& origin, const point& dir,
                     double level) {
    point p = origin;
    double v;
    do {
        advance(p, dir);
        v = f(p);
    } while (v < level);
    return p;
}

void push_tet(std::vector<double>& soup, const point& a, const point& b, const point& c, const point& d) {
    for (const point* p : {&a, &b, &c, &d})
        for (double x : *p) soup.push_back(x);
}

}  // namespace

void object3d_base::init_polar(const std::function<double(std::array<double, 3>)>& potential_function, double x0,
                               double y0, double z0, double level_value, double step, double angle_step,
                               tetra_type arg_tetra_type, double tetra_v1, double /*tetra_v2*/) {
    if (arg_tetra_type != tetra_type::solid)
        throw std::runtime_error("init_polar: only solid objects are generated procedurally");
    const point ray_step{0.001, 0, 0};  // hard-coded in the reference (object3d_base.cpp:87), `step` is only used along z
    const double d_angle = PI / angle_step;
    const point centre{x0, y0, z0};

    const point top = march_to_level(potential_function, centre, point{0, 0, step}, level_value);
    const point bottom = march_to_level(potential_function, centre, point{0, 0, -step}, level_value);

    // rings of surface points; both angles are accumulated sums (object3d_base.cpp:92-93,123-143), which
    // decides how many rings/points exist: 255 x 256 for angle_step 128, 511 x 511 for 256
    std::vector<std::vector<point>> ring;
    double angle_x = 0;
    double angle_y = -PI + d_angle;
    while (angle_y < (PI - d_angle + std::numeric_limits<double>::epsilon())) {
        const point tilted = turn_about_z(ray_step, angle_y);
        angle_y += d_angle;
        std::vector<point> pts;
        while (angle_x < 2 * PI - d_angle + std::numeric_limits<double>::epsilon()) {
            pts.push_back(march_to_level(potential_function, centre, turn_about_y(tilted, angle_x), level_value));
            angle_x += d_angle;
        }
        ring.push_back(std::move(pts));
        angle_x = 0;
    }

    // centre-fan tetrahedra (object3d_base.cpp:152-193)
    std::vector<double> soup;
    const std::size_t n = ring[0].size();
    const std::size_t last = ring.size() - 1;
    soup.reserve(12 * (2 * n + last * 2 * n));
    for (std::size_t i = 1; i < n; ++i) push_tet(soup, centre, bottom, ring[0][i], ring[0][i - 1]);
    push_tet(soup, centre, bottom, ring[0][0], ring[0][n - 1]);
    for (std::size_t i = 1; i < n; ++i) push_tet(soup, centre, top, ring[last][i], ring[last][i - 1]);
    push_tet(soup, centre, top, ring[last][0], ring[0][n - 1]);  // sic: closes with ring 0 (object3d_base.cpp:171-174)
    for (std::size_t i = 1; i <= last; ++i) {
        for (std::size_t j = 1; j < n; ++j) {
            push_tet(soup, centre, ring[i - 1][j - 1], ring[i - 1][j], ring[i][j - 1]);
            push_tet(soup, centre, ring[i][j - 1], ring[i][j], ring[i - 1][j]);
        }
        push_tet(soup, centre, ring[i - 1][n - 1], ring[i - 1][0], ring[i][n - 1]);
        push_tet(soup, centre, ring[i][n - 1], ring[i][0], ring[i - 1][0]);
    }

    object3d_data& d = *_data;
    d.kind = tetra_type::solid;
    d.soup = std::move(soup);
    d.colour = tetra_v1;
    d.points.clear();
    d.cells.clear();
}

// ---------------------------------------------------------------------------------------------
// concrete objects
// ---------------------------------------------------------------------------------------------
object3d_accretion_disk::object3d_accretion_disk(const std::string& filename) {
    read_vtk_file(filename, {"AbsorpCoef", "radEnLooseRate"});  // object3d_accretion_disk.cpp:4
}

namespace {
double norm2(const point& v) {  // object3d_roche_lobe.cpp:3-9
    double s{};
    for (double x : v) s += x * x;
    return sqrt(s);
}
point cross(const point& a, const point& b) {  // object3d_roche_lobe.cpp:11-18
    return point{a[1] * b[2] - a[2] * b[1], -(a[0] * b[2]) + (a[2] * b[0]), a[0] * b[1] - a[1] * b[0]};
}
}  // namespace

object3d_roche_lobe::object3d_roche_lobe(const point& pos_accretor, double dist, double donor_angle_around_y,
                                         double m_accretor, double m_donor, double def_omega) {
    const double donor_x = pos_accretor[0] - dist;
    const double mass_centre_x = (donor_x * m_donor + pos_accretor[0] * m_accretor) / (m_accretor + m_donor);
    // The reference computes the analytic L1 estimate and then overrides it (object3d_roche_lobe.cpp:25-30).
    const double lagrange1_x = 0.35515;

    auto potential = [&](const std::array<double, 3>& r) -> double {
        const double to_accretor = norm2({r[0] - pos_accretor[0], r[1], r[2]});
        const double to_donor = norm2({r[0] - donor_x, r[1], r[2]});
        const double spin = norm2(cross({r[0] - mass_centre_x, r[1], r[2]}, {0, def_omega, 0}));
        const double centrifugal = (1. / 2.) * spin * spin;
        // long double arithmetic through G_SOL, rounded to double once (object3d_roche_lobe.cpp:42)
        double F = -((G_SOL * m_accretor) / to_accretor) - ((G_SOL * m_donor) / to_donor) - centrifugal;
        return F;
    };

    const double level = potential({lagrange1_x, 0, 0});
    init_polar(potential, donor_x, 0, 0, level, 0.001, 128, tetra_type::solid,
               app::instance().config.roche_lobe_solid_color);
    rotate_around_y_axis(donor_angle_around_y, ACC_X0);  // object3d_roche_lobe.cpp:48
}

object3d_sphere::object3d_sphere(const point& center, double R) {
    auto radius = [&](const point& p) { return norm2(point{p[0] - center[0], p[1] - center[1], p[2] - center[2]}); };
    init_polar(radius, center[0], center[1], center[2], R, 0.001, 256, tetra_type::solid,
               app::instance().config.acc_disk_solid_color);
}
