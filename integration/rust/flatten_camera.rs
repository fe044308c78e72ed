// Append to src/cam.rs (Camera's fields are private; `w` is dead in the reference, cam.rs:13-14).
impl Camera {
    pub fn to_gpu(&self) -> crate::gpu::RtCamera {
        let a = |v: &Vector| [v[0], v[1], v[2]];
        crate::gpu::RtCamera {
            origin: a(&self.origin),
            lower_left: a(&self.lower_left_corner),
            horizontal: a(&self.horizontal),
            vertical: a(&self.vertical),
            u: a(&self.u),
            v: a(&self.v),
            lens_radius: self.lens_radius,
            time_open: self.time_open,
            time_close: self.time_close,
        }
    }
}
