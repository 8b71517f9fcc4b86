// camera.hpp — host-side pinhole camera state; mirrors raytracer_lib/src/scene/camera.rs.
// The device only ever sees the derived rotation matrix, origin and max_x/max_y.
#pragma once
#include <cstdint>
#include "vecmath.hpp"

namespace mi355rt {

struct Ray { Vec3 pos, dir; };

class Camera {
public:
    Camera() = default;
    // camera.rs:22-61 — only rotation and position are expected in the matrix
    static Camera from_orientation_matrix(uint32_t width, uint32_t height, const Matrix& orientation, float fov_deg)
    {
        Camera c;
        Matrix rot = orientation;
        rot.e[3] = 0.0f; rot.e[7] = 0.0f; rot.e[11] = 0.0f;
        rot.e[12] = 0.0f; rot.e[13] = 0.0f; rot.e[14] = 0.0f; rot.e[15] = 1.0f;
        float fov = fov_deg * 3.14159274101257324f / 180.0f;
        float half_fov = 0.5f * fov;
        c.max_x_ = 1.0f * std::tan(half_fov);
        c.max_y_ = 1.0f * std::tan(half_fov);     // aspect ratio is ignored, camera.rs:41-44
        c.width_ = width; c.height_ = height;
        c.base_orientation_ = orientation;
        c.base_rotation_ = rot;
        c.update_matrices();
        return c;
    }
    void add_x_angle(float r) { x_angle_ += r; update_matrices(); }           // camera.rs:63-66
    void add_y_angle(float r) { y_angle_ += r; update_matrices(); }           // camera.rs:68-71
    void move_rel(float x, float y, float z)                                   // camera.rs:73-78
    {
        pos_.x += x; pos_.y += y; pos_.z += z;
        update_matrices();
    }
    // camera.rs:80-90 with the two random draws passed in
    Ray get_ray(uint32_t u, uint32_t v, float xi1, float xi2) const
    {
        float dir_x = -max_x_ + 2.0f * max_x_ * (((float)u + xi1) / (float)width_);
        float dir_y = -max_y_ + 2.0f * max_y_ * (((float)v + xi2) / (float)height_);
        Vec4 dir = rotation_ * Vec4(dir_x, -dir_y, 1.0f, 1.0f);
        Vec4 pos = orientation_ * Vec4(0.0f, 0.0f, 0.0f, 1.0f);
        return Ray{ pos.xyz(), dir.xyz() };
    }
    const Matrix& rotation() const { return rotation_; }
    const Matrix& orientation() const { return orientation_; }
    float max_x() const { return max_x_; }
    float max_y() const { return max_y_; }
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }

private:
    void update_matrices()                                                     // camera.rs:92-98
    {
        rotation_ = Matrix::rot_x(x_angle_) * Matrix::rot_y(y_angle_) * base_rotation_;
        orientation_ = rotation_ * Matrix::translate(pos_) * base_orientation_;
    }
    float x_angle_ = 0.0f, y_angle_ = 0.0f;
    Vec3 pos_;
    uint32_t width_ = 0, height_ = 0;
    Matrix base_orientation_ = Matrix::ident(), base_rotation_ = Matrix::ident();
    Matrix orientation_ = Matrix::ident(), rotation_ = Matrix::ident();
    float max_x_ = 0.0f, max_y_ = 0.0f;
};

}  // namespace mi355rt
