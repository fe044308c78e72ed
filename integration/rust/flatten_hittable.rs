// Append to src/hittable.rs (the fields of Cube, Sphere, ... are private to this module).
// C++ twin, compiled and tested in this repository: raytracer_amd/csrc/host/rth_scene.cpp (Flattener).
use crate::gpu::{self, FlatScene};

impl Hittable {
    /// Pre-order flattening of the tree `new_bvh` built (hittable.rs:159-206): a node's left child node, if
    /// any, is the node's index + 1.  Returns (reference, height in nodes).
    pub fn flatten_into(&self, fs: &mut FlatScene) -> Result<(u32, u32), String> {
        match self {
            Hittable::Bvh(node) => {
                let me = fs.node_left.len();
                let (lo, hi) = (&node.bounding_box.minimum, &node.bounding_box.maximum);
                fs.node_box.extend_from_slice(&[lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]]);
                fs.node_left.push(gpu::RT_REF_NONE);
                fs.node_right.push(gpu::RT_REF_NONE);
                let (l, hl) = node.left.flatten_into(fs)?;
                let (r, hr) = node.right.flatten_into(fs)?;
                fs.node_left[me] = l;
                fs.node_right[me] = r;
                Ok((me as u32, 1 + hl.max(hr)))
            }
            Hittable::Empty => Ok((gpu::RT_REF_NONE, 0)), // only as the whole world (hittable.rs:171)
            Hittable::List(_) => Err("a List inside the world tree is not representable".into()),
            leaf => {
                // inside an instance's sub-tree every leaf carries the id of the world object that owns the instance
                let object = match fs.instance_owner {
                    Some(owner) => owner,
                    None => {
                        fs.next_object += 1;
                        fs.next_object - 1
                    }
                };
                Ok((leaf.flatten_object(fs, object)?, 0))
            }
        }
    }

    /// The whole world: its tree first, then — behind it in the node arrays, so that the world's own nodes stay in
    /// pre-order — the sub-tree of every instance (Translate / Rotate over a Bvh, hittable.rs:81-91 over 93-98; rt.h,
    /// ABI version 3): the complex's `body` becomes the INDEX OF THE NODE that roots its sub-tree.  Returns
    /// (root reference, height in nodes: the world's + 1 + the tallest sub-tree's, what rt_scene_create checks
    /// against RT_MAX_BVH_HEIGHT).
    pub fn flatten_world(&self, fs: &mut FlatScene) -> Result<(u32, u32), String> {
        let (root, height) = self.flatten_into(fs)?;
        let mut tallest = 0;
        for (complex, owner, sub) in std::mem::take(&mut fs.pending_instances) {
            fs.instance_owner = Some(owner);
            let (sub_root, h) = unsafe { &*sub }.flatten_into(fs)?; // its leaves must be bare primitives
            fs.instance_owner = None;
            if sub_root & gpu::RT_REF_LEAF != 0 {
                return Err("an instance of a single object: wrap the object itself".into());
            }
            fs.complex_body[complex as usize] = sub_root;
            tallest = tallest.max(h);
        }
        Ok((root, if tallest > 0 { height + 1 + tallest } else { height }))
    }

    /// One world object: a primitive under a chain of Translate / Rotate / ConstantMedium wrappers,
    /// recorded outermost first (hittable.rs:319-386).
    fn flatten_object(&self, fs: &mut FlatScene, object: u32) -> Result<u32, String> {
        let mut ops: Vec<(u8, [f64; 4])> = Vec::new();
        let mut medium_material = gpu::RT_REF_NONE;
        let mut cur = self;
        loop {
            match cur {
                Hittable::Translate(t) => {
                    ops.push((gpu::OP_TRANSLATE, [t.offset[0], t.offset[1], t.offset[2], 0.0]));
                    cur = &t.hittable;
                }
                Hittable::Rotate(r) => {
                    let q = &r.rotation; // (w, x, y, z), alias.rs:130-133
                    ops.push((gpu::OP_ROTATE, [q[0], q[1], q[2], q[3]]));
                    cur = &r.hittable;
                }
                Hittable::ConstantMedium(m) => {
                    if medium_material != gpu::RT_REF_NONE {
                        return Err("a constant medium inside a constant medium is not representable".into());
                    }
                    medium_material = fs.add_material(&m.material); // Isotropic, hittable.rs:148
                    ops.push((gpu::OP_MEDIUM, [m.neg_inv_density, 0.0, 0.0, 0.0]));
                    cur = &m.boundary;
                }
                _ => break,
            }
        }
        let body = match cur {
            Hittable::Sphere(s) => {
                let mat = fs.add_material(&s.material);
                let i = fs.sphere_material.len() as u32;
                fs.sphere.extend_from_slice(&[s.center[0], s.center[1], s.center[2], s.radius]);
                fs.sphere_material.push(mat);
                fs.sphere_object.push(object);
                gpu::make_leaf(gpu::LEAF_SPHERE, i)
            }
            Hittable::MovingSphere(s) => {
                let mat = fs.add_material(&s.material);
                let i = fs.msphere_material.len() as u32;
                fs.msphere.extend_from_slice(&[
                    s.center_start[0], s.center_start[1], s.center_start[2],
                    s.center_end[0], s.center_end[1], s.center_end[2],
                    s.radius, s.time_start, s.time_end,
                ]);
                fs.msphere_material.push(mat);
                fs.msphere_object.push(object);
                gpu::make_leaf(gpu::LEAF_MSPHERE, i)
            }
            Hittable::Cube(c) => {
                let mat = fs.add_material(&c.material);
                let i = fs.cube_material.len() as u32;
                fs.cube.extend_from_slice(&[c.p0[0], c.p0[1], c.p0[2], c.p1[0], c.p1[1], c.p1[2]]);
                fs.cube_material.push(mat);
                fs.cube_object.push(object);
                gpu::make_leaf(gpu::LEAF_CUBE, i)
            }
            Hittable::Bvh(_) if !ops.is_empty() && medium_material == gpu::RT_REF_NONE && fs.instance_owner.is_none() => {
                // an INSTANCE: flatten_world() fills in the node index once the world's own tree is laid out
                fs.pending_instances.push((fs.complex_body.len() as u32, object, cur as *const Hittable));
                gpu::RT_REF_NONE
            }
            _ => return Err("a List or Empty under a wrapper, a medium over a Bvh, or nested instances: not representable".into()),
        };
        if ops.is_empty() {
            return Ok(body);
        }
        let me = fs.complex_body.len() as u32;
        fs.complex_body.push(body);
        fs.complex_op_first.push(fs.op_tag.len() as u32);
        fs.complex_op_count.push(ops.len() as u32);
        fs.complex_material.push(medium_material);
        fs.complex_object.push(object);
        for (tag, data) in ops {
            fs.op_tag.push(tag);
            fs.op_data.extend_from_slice(&data);
        }
        Ok(gpu::make_leaf(gpu::LEAF_COMPLEX, me))
    }

    /// One entry of `lights` (hittable.rs:477-515: only a bare Sphere is importance-sampled; anything else
    /// has pdf_value 0 and random() = (1, 0, 0)).
    pub fn push_light(&self, fs: &mut FlatScene) {
        match self {
            Hittable::Sphere(s) => {
                fs.light_tag.push(gpu::LIGHT_SPHERE);
                fs.light_sphere.extend_from_slice(&[s.center[0], s.center[1], s.center[2], s.radius]);
            }
            _ => {
                fs.light_tag.push(gpu::LIGHT_OTHER);
                fs.light_sphere.extend_from_slice(&[0.0; 4]);
            }
        }
    }
}
