// Shipped XML scenes through the headless SimContext (host/SimContext.hpp), with the setup() step each
// sample main performs between load() and initialize():
//   flag / flag_nowind  samples/windyflag/windyflag.cpp:68-128   two StaticAnchors (0, <length>), WindForce (10,0,2) over all dynamic faces
//   plinko              samples/plinkopony/plinkopony.cpp:53-96  one CollisionCylinder per object named c*, one CollisionForce
//   scale1.3            deterministic stand-in for samples/bunnyexpand/bunnyexpand.cpp:45-63 (x *= 1.3 after initialize)
// usage: scene_run <xml> <setup> <dump.bin> <frames> [traj.bin]
//   frames < 0: host only -- load(), setup, add_scene_forces(), dump; no device is touched.
// dump.bin: int32 dof, n_forces, n_explicit, iters, n_objects; f64 dt; f64 x[dof], m[dof];
//           per force int32 kind, idx[4], f64 par[4]; per explicit int32 type, n_idx, f64 dir[3], int32 idx[n_idx];
//           object names in object_params order, '\n' separated, preceded by int32 byte count;
//           collision cylinders: int32 n, f64 (cx,cy,cz,r)[n]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "SimContext.hpp"
using namespace admm;

static void put_i(FILE *f, int v) { fwrite(&v, 4, 1, f); }
static void put_d(FILE *f, double v) { fwrite(&v, 8, 1, f); }

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: scene_run <xml> <setup> <dump.bin> <frames> [traj.bin]\n"); return 1; }
    const std::string xml = argv[1], setup = argv[2];
    const int frames = atoi(argv[4]);
    std::vector<std::shared_ptr<CollisionShape> > shapes;
    try {
        SimContext context;
        context.system->settings.verbose = 0;
        context.load(xml);
        context.system->settings.verbose = 0;

        if (setup == "flag" || setup == "flag_nowind") {
            std::vector<mcl::Param> cloth_params = context.scene->object_params["cloth1"];
            int cloth_height = 0;
            for (size_t i = 0; i < cloth_params.size(); ++i) if (cloth_params[i].tag == "length") cloth_height = cloth_params[i].as_int();
            context.system->forces.push_back(std::shared_ptr<Force>(new StaticAnchor(0)));
            context.system->forces.push_back(std::shared_ptr<Force>(new StaticAnchor(cloth_height)));
            if (setup == "flag") {
                std::vector<int> faces = context.dynamic_faces();
                std::shared_ptr<ExplicitForce> wind(new WindForce(faces));
                wind->direction = Vector3d(10, 0, 2);
                context.system->explicit_forces.push_back(wind);
            }
        } else if (setup == "plinko") {
            std::unordered_map<std::string, std::vector<mcl::Param> >::iterator it = context.scene->object_params.begin();
            for (; it != context.scene->object_params.end(); ++it) {
                if (it->first[0] != 'c') continue;
                double rad = 1.f;
                Vector3d center(0, 0, 0), scale(1, 1, 1);
                for (size_t i = 0; i < it->second.size(); ++i) {
                    if (it->second[i].tag == "scale_copy") { trimesh::vec v = it->second[i].as_vec3(); scale = Vector3d(v[0], v[1], v[2]); }
                    else if (it->second[i].tag == "translate_copy") { trimesh::vec v = it->second[i].as_vec3(); center = Vector3d(v[0], v[1], v[2]); }
                    else if (it->second[i].tag == "radius") rad = it->second[i].as_double();
                }
                shapes.push_back(std::shared_ptr<CollisionShape>(new CollisionCylinder(center, scale, rad)));
            }
            context.system->forces.push_back(std::shared_ptr<Force>(new CollisionForce(shapes)));
        }

        if (frames < 0) context.add_scene_forces();
        else context.initialize();

        System &S = *context.system;
        FILE *f = fopen(argv[3], "wb");
        if (!f) return 1;
        const int dof = (int)S.m_x.size();
        put_i(f, dof); put_i(f, (int)S.forces.size()); put_i(f, (int)S.explicit_forces.size()); put_i(f, S.settings.admm_iters); put_i(f, (int)context.scene->object_params.size());
        put_d(f, S.settings.timestep_s);
        fwrite(S.m_x.data(), 8, dof, f); fwrite(S.m_masses.data(), 8, dof, f);
        for (size_t i = 0; i < S.forces.size(); ++i) {
            int idx[4] = {0, 0, 0, 0}; double par[4] = {0, 0, 0, 0};
            const int kind = S.forces[i]->kind();
            if (kind == ADMM_KIND_COLLISION) { idx[0] = (int)static_cast<CollisionForce *>(S.forces[i].get())->collisionShapes.size(); par[0] = S.forces[i]->weight; }
            else S.forces[i]->describe(idx, par);
            put_i(f, kind); fwrite(idx, 4, 4, f); fwrite(par, 8, 4, f);
        }
        for (size_t i = 0; i < S.explicit_forces.size(); ++i) {
            const ExplicitForce &e = *S.explicit_forces[i];
            const std::vector<int> &l = e.index_list();
            put_i(f, e.explicit_type()); put_i(f, (int)l.size());
            for (int j = 0; j < 3; ++j) put_d(f, e.direction[j]);
            if (!l.empty()) fwrite(l.data(), 4, l.size(), f);
        }
        std::string names;
        for (std::unordered_map<std::string, std::vector<mcl::Param> >::iterator it = context.scene->object_params.begin(); it != context.scene->object_params.end(); ++it) names += it->first + "\n";
        put_i(f, (int)names.size()); fwrite(names.data(), 1, names.size(), f);
        put_i(f, (int)shapes.size());
        for (size_t s = 0; s < shapes.size(); ++s) { for (int j = 0; j < 3; ++j) put_d(f, shapes[s]->center[j]); put_d(f, shapes[s]->shape_radius()); }
        // surface / triangle faces of every dynamic object, in object_params order
        for (std::unordered_map<std::string, std::vector<mcl::Param> >::iterator it = context.scene->object_params.begin(); it != context.scene->object_params.end(); ++it) {
            std::shared_ptr<trimesh::TriMesh> mesh = context.scene->objects_map[it->first]->get_TriMesh();
            if (!mesh) { put_i(f, -1); put_i(f, -1); continue; }
            put_i(f, (int)mesh->vertices.size()); put_i(f, (int)mesh->faces.size());
            for (size_t k = 0; k < mesh->faces.size(); ++k) fwrite(mesh->faces[k].v, 4, 3, f);
        }
        fclose(f);
        if (frames < 0) return 0;

        if (setup == "scale1.3") for (int i = 0; i < dof; ++i) S.m_x[i] = S.m_x[i] * 1.3;
        FILE *t = argc > 5 ? fopen(argv[5], "wb") : 0;
        for (int fr = 0; fr < frames; ++fr) {
            if (!context.step()) return 3;
            if (t) fwrite(S.m_x.data(), 8, dof, t);
        }
        if (t) fclose(t);
        context.update();
    } catch (std::exception &e) {
        fprintf(stderr, "scene_run: %s\n", e.what());
        return 2;
    }
    return 0;
}
