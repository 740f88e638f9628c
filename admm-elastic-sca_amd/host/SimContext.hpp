// SimContext.hpp -- headless mirror of the reference's scene ingest layer:
// `SimContext` (src/SimContext.{hpp,cpp}) and `admm::ForceBuilder`
// (src/ForceBuilder.{hpp,cpp}).  Same public surface the sample mains use
// (context->load / initialize / step / update, context->system, context->scene,
// settings.run_realtime), so samples/*/*.cpp port by deleting their GL lines;
// the admm::System underneath is the HIP-backed one (admm/System.hpp).
//
// What must match the reference bit for bit, and is tested against fixtures dumped
// from the compiled reference loader (tests/golden/scene_*.npz):
//   * node positions (float mesh vertices widened to double) and their order,
//   * node masses: uniform (mass / #vertices) or density weighted, accumulated tet by tet /
//     face by face in mesh order (ForceBuilder.hpp:191-297),
//   * the force list: one force per element per <Force> parameter of the object, in parameter
//     order then element order; bend hinges in the across-edge walk order with first-seen
//     de-duplication (ForceBuilder.cpp:76-278); springs per first-seen edge (:226-264),
//   * explicit forces / anchors / wind appended by initialize() in the iteration order of the
//     name -> parameters hash table (SimContext.cpp:104-161); std::unordered_map is used here
//     too, so the order is the reference's for the same libstdc++.
//
// Differences: no statics (one ForceBuilder per context, so several contexts can coexist),
// errors that the reference turns into assert(false)/exit(0) are std::runtime_error here.
#pragma once
#include <set>
#include <utility>

#include "MCL/Scene.hpp"
#include "admm/System.hpp"

namespace admm {

class ForceBuilder {
public:
    ForceBuilder() : index_offset(0), num_objects(0), bend_index(0), force_param_map(0), system_to_scene_map(0) {}

    std::shared_ptr<admm::System> system;
    int index_offset;   // first system node of the object being built
    int num_objects;
    int bend_index;
    std::unordered_map<std::string, mcl::Component> *force_param_map;
    std::unordered_map<int, std::pair<int, int> > *system_to_scene_map; // system node -> (object, vertex)

    // ForceBuilder.cpp:76-278
    static bool build_trimesh(std::shared_ptr<trimesh::TriMesh> mesh, mcl::Component &force,
                              std::vector<std::shared_ptr<Force> > *sys_forces, int idx_offset, int *bend_counter = 0) {
        const std::string force_type = mcl::parse::to_lower(force.type);
        trimesh::TriMesh &m = *mesh;
        std::set<std::vector<int> > hinge_seen;          // sorted 4-tuples already emitted
        std::set<std::pair<int, int> > edge_seen;        // sorted node pairs already emitted
        for (size_t f = 0; f < m.faces.size(); ++f) {
            const int p[3] = {m.faces[f][0] + idx_offset, m.faces[f][1] + idx_offset, m.faces[f][2] + idx_offset};
            if (force_type == "lineartrianglestrain" || force_type == "trianglestrain") {
                trimesh::vec2 limit(0.f, 9999999.f);
                if (force.exists("limit")) limit = force["limit"].as_vec2();
                if (!need(force, "stiffness")) return false;
                const double stiffness = force["stiffness"].as_double();
                sys_forces->push_back(std::shared_ptr<Force>(new LimitedTriangleStrain(p[0], p[1], p[2], stiffness, limit[0], limit[1])));
            } else if (force_type == "bend") {
                if (!need(force, "stiffness")) return false;
                const double stiffness = force["stiffness"].as_double();
                if (m.across_edge.size() != m.faces.size()) m.need_across_edge();
                // corner c faces neighbour g across edge (c+1, c+2): hinge = (p_c, far vertex of g, p_{c+2}, p_{c+1})
                for (int c = 0; c < 3; ++c) {
                    const int g = m.across_edge[f][c];
                    if (g < 0) continue;
                    int hv[4] = {p[c], far_vertex(m, g, (int)f) + idx_offset, p[(c + 2) % 3], p[(c + 1) % 3]};
                    std::vector<int> key(hv, hv + 4);
                    std::sort(key.begin(), key.end());
                    if (!hinge_seen.insert(key).second) continue;
                    sys_forces->push_back(std::shared_ptr<Force>(new BendForce(hv[0], hv[1], hv[2], hv[3], stiffness)));
                    if (bend_counter) *bend_counter += 1;
                }
            } else if (force_type == "spring") {
                const int e[3][2] = {{p[0], p[1]}, {p[0], p[2]}, {p[1], p[2]}};
                for (int k = 0; k < 3; ++k) {
                    if (!edge_seen.insert(std::make_pair(std::min(e[k][0], e[k][1]), std::max(e[k][0], e[k][1]))).second) continue;
                    trimesh::vec2 limit(-1.f, -1.f);
                    if (force.exists("limit")) limit = force["limit"].as_vec2();
                    if (!need(force, "stiffness")) return false;
                    const double stiffness = force["stiffness"].as_double();
                    if (limit[0] >= 0.f) { std::cout << "TODO: ForceBuilder::build_trimesh with limited springs" << std::endl; return false; }
                    sys_forces->push_back(std::shared_ptr<Force>(new Spring(e[k][0], e[k][1], stiffness)));
                }
            } else if (force_type != "constforce") {
                std::cout << "TODO: ForceBuilder::build_trimesh with force: " << force_type << std::endl;
                return false;
            }
        }
        return true;
    }

    // ForceBuilder.cpp:281-446
    static bool build_tetmesh(std::shared_ptr<mcl::TetMesh> mesh, mcl::Component &force,
                              std::vector<std::shared_ptr<Force> > *sys_forces, int idx_offset) {
        const std::string force_type = mcl::parse::to_lower(force.type);
        for (size_t t = 0; t < mesh->tets.size(); ++t) {
            int p[4];
            for (int j = 0; j < 4; ++j) p[j] = mesh->tets[t].v[j] + idx_offset;
            if (force_type == "lineartetstrain") {
                if (!need(force, "stiffness")) return false;
                const double stiffness = force["stiffness"].as_double();
                double weight_scale = 1.0;
                if (force.exists("weight_scale")) weight_scale = force["weight_scale"].as_double();
                sys_forces->push_back(std::shared_ptr<Force>(new LinearTetStrain(p[0], p[1], p[2], p[3], stiffness, weight_scale)));
            } else if (force_type == "neohookeantet" || force_type == "stvktet") {
                if (!force.exists("mu") || !force.exists("lambda")) {
                    std::cerr << "\n**ForceBuilder Error: force \"" << force.name << "\" needs mu and lambda parameters." << std::endl;
                    return false;
                }
                const double mu = force["mu"].as_double(), lambda = force["lambda"].as_double();
                int max_iters = 10;
                if (force.exists("max_iterations")) max_iters = force["max_iterations"].as_int();
                sys_forces->push_back(std::shared_ptr<Force>(new HyperElasticTet(p[0], p[1], p[2], p[3], mu, lambda, max_iters, force_type == "stvktet" ? "stvk" : "nh")));
            } else if (force_type == "volpres") {
                if (!need(force, "stiffness") || !need(force, "range_min") || !need(force, "range_max")) return false;
                sys_forces->push_back(std::shared_ptr<Force>(new TetVolume(p[0], p[1], p[2], p[3], force["stiffness"].as_double(),
                                                                           force["range_min"].as_double(), force["range_max"].as_double())));
            } else if (force_type != "constforce") {
                std::cout << "TODO: ForceBuilder::build_tetmesh with force: " << force_type << std::endl;
                return false;
            }
        }
        return true;
    }

    // The SceneManager's object callback (ForceBuilder.hpp:75-309): build the mesh, then -- if the object
    // names a <Force> -- append its vertices as system nodes, its elements as forces, and lump its mass.
    std::shared_ptr<mcl::BaseObject> admm_build_object(mcl::Component &obj) {
        num_objects++;
        const std::string o_type = mcl::parse::to_lower(obj.type);
        std::shared_ptr<mcl::BaseObject> object = mcl::default_build_object(obj);
        if (!obj.exists("force")) return object; // static scenery
        std::shared_ptr<trimesh::TriMesh> mesh = object->get_TriMesh();
        if (!mesh) throw std::runtime_error("\n**ForceBuilder Error: object \"" + obj.name + "\" of type \"" + obj.type +
                                            "\" has a Force, but this headless loader builds geometry only for tetmesh, plane, sphere, box, beam, cylinder and torus objects and for OBJ / PLY meshes (not for other mesh files or point clouds)");
        double objMass = -1.0;
        if (obj.exists("mass")) objMass = obj.get("mass").as_double();
        if (objMass < 0.0) throw std::runtime_error("\n**Error: You must specify mass (kg) for object " + obj.name + ", e.g. <Mass type=\"double\" value=\"2\" />");
        const int nv = (int)mesh->vertices.size();
        const double node_mass = objMass / mesh->vertices.size();
        bool density_weighted_mass = true;
        if (obj.exists("density_weighted_mass")) density_weighted_mass = obj.get("density_weighted_mass").as_bool();

        VectorXd &X = system->m_x, &M = system->m_masses;
        const int old_nodes = (int)X.size() / 3;
        X.conservativeResize((old_nodes + nv) * 3);
        system->m_v.conservativeResize((old_nodes + nv) * 3);
        M.conservativeResize((old_nodes + nv) * 3);
        for (int i = 0; i < nv; ++i) {
            const int s = old_nodes + i;
            if (system_to_scene_map) system_to_scene_map->insert(std::make_pair(s, std::make_pair(num_objects - 1, i)));
            const trimesh::point q = mesh->vertices[i];
            for (int j = 0; j < 3; ++j) { X[3 * s + j] = q[j]; system->m_v[3 * s + j] = 0.0; M[3 * s + j] = density_weighted_mass ? 0.0 : node_mass; }
        }

        for (size_t i = 0; i < obj.params.size(); ++i) {
            if (mcl::parse::to_lower(obj.params[i].tag) != "force") continue;
            const std::string f_name = obj.params[i].value;
            if (!force_param_map || force_param_map->count(f_name) == 0)
                throw std::runtime_error("\n**ForceBuilder::Error: No force named \"" + f_name + "\" for object \"" + obj.name + "\"");
            mcl::Component force = force_param_map->at(f_name);
            bool ok;
            if (o_type == "tetmesh") {
                std::shared_ptr<mcl::TetMesh> t_mesh = std::static_pointer_cast<mcl::TetMesh>(object);
                if (system->settings.verbose > 0) std::cout << "Tetmesh " << obj.name << " has " << t_mesh->tets.size() << " tets." << std::endl;
                ok = build_tetmesh(t_mesh, force, &system->forces, index_offset);
            } else {
                if (system->settings.verbose > 0) std::cout << "Trimesh " << obj.name << " has " << mesh->faces.size() << " tris." << std::endl;
                ok = build_trimesh(mesh, force, &system->forces, index_offset, &bend_index);
            }
            (void)ok; // like the reference, a force type the builder does not know is reported and skipped
        }

        if (density_weighted_mass) {
            if (o_type == "tetmesh") {
                std::shared_ptr<mcl::TetMesh> t_mesh = std::static_pointer_cast<mcl::TetMesh>(object);
                double total = 0;
                for (size_t t = 0; t < t_mesh->tets.size(); ++t) total += tet_volume(X, t_mesh->tets[t].v, index_offset);
                if (!(total > 0)) throw std::runtime_error("\n**Error: tet object volume is zero, so can't compute mass density.");
                const double density = objMass / total;
                for (size_t t = 0; t < t_mesh->tets.size(); ++t) {
                    const double tetMass = density * tet_volume(X, t_mesh->tets[t].v, index_offset);
                    for (int j = 0; j < 4; ++j) { const int n = t_mesh->tets[t].v[j] + index_offset; for (int k = 0; k < 3; ++k) M[3 * n + k] += tetMass / 4.0; }
                }
            } else {
                double total = 0;
                for (size_t f = 0; f < mesh->faces.size(); ++f) total += tri_area(X, mesh->faces[f].v, index_offset);
                if (!(total > 0)) throw std::runtime_error("\n**Error: tri object area is zero, so can't compute mass density.");
                const double density = objMass / total;
                for (size_t f = 0; f < mesh->faces.size(); ++f) {
                    const double triMass = density * tri_area(X, mesh->faces[f].v, index_offset);
                    for (int j = 0; j < 3; ++j) { const int n = mesh->faces[f][j] + index_offset; for (int k = 0; k < 3; ++k) M[3 * n + k] += triMass / 3.0; }
                }
            }
        }
        index_offset += nv;
        return object;
    }

private:
    static bool need(mcl::Component &force, const char *tag) {
        if (force.exists(tag)) return true;
        std::cerr << "\n**ForceBuilder Error: force \"" << force.name << "\" needs a " << tag << " parameter." << std::endl;
        return false;
    }
    // the vertex of face `g` that face `f` does not have (ForceBuilder.cpp:25-54)
    static int far_vertex(const trimesh::TriMesh &m, int g, int f) {
        int shared = 0, lone = -1;
        for (int i = 0; i < 3; ++i) {
            if (m.faces[f].indexof(m.faces[g][i]) >= 0) shared++;
            else if (lone < 0) lone = m.faces[g][i];
        }
        if (shared != 2) throw std::runtime_error("Error in getUniqueVert: two input faces do not share 2 verts!");
        return lone;
    }
    // fixed-size Eigen expressions: dot = a0 b0 + (a1 b1 + a2 b2)
    struct P3 { double x, y, z; };
    static P3 at(const VectorXd &X, int n) { P3 r = {X[3 * n], X[3 * n + 1], X[3 * n + 2]}; return r; }
    static P3 sub(const P3 &a, const P3 &b) { P3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
    static P3 cross(const P3 &a, const P3 &b) { P3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; }
    static double dot(const P3 &a, const P3 &b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
    static double tet_volume(const VectorXd &X, const int *v, int off) {
        const P3 v0 = at(X, v[0] + off), v1 = at(X, v[1] + off), v2 = at(X, v[2] + off), v3 = at(X, v[3] + off);
        return std::fabs(dot(sub(v0, v3), cross(sub(v1, v3), sub(v2, v3)))) / 6.0;
    }
    static double tri_area(const VectorXd &X, const int *v, int off) {
        const P3 v0 = at(X, v[0] + off), v1 = at(X, v[1] + off), v2 = at(X, v[2] + off);
        const P3 c = cross(sub(v1, v0), sub(v2, v0));
        return 0.5 * std::sqrt(dot(c, c));
    }
};

} // namespace admm

class SimContext {
public:
    struct Settings {
        bool run_realtime; // <realtime value="1" />
        Settings() : run_realtime(false) {}
    } settings;

    std::shared_ptr<admm::System> system;
    std::shared_ptr<mcl::SceneManager> scene;

    SimContext() : system(new admm::System()), scene(new mcl::SceneManager()) {
        builder.system = system;
        builder.system_to_scene_map = &system_to_scene_map;
        builder.force_param_map = &force_param_map;
        scene->createObject = [this](mcl::Component &c) { return builder.admm_build_object(c); };
    }

    // SimContext.cpp:39-100.  <admmelastic> is read first (solver settings, named force parameter sets),
    // then <mclScene>, whose objects pull the named forces in.  Throws std::runtime_error.
    void load(std::string config_file) {
        mcl::xml::Node doc;
        if (!mcl::xml::load_file(config_file, doc)) throw std::runtime_error("\n**SimContext::load Error: Unable to load " + config_file);
        const mcl::xml::Node *head = mcl::xml::find_head(doc, "admmelastic");
        for (size_t c = 0; head && c < head->children.size(); ++c) {
            const mcl::xml::Node &n = head->children[c];
            const std::string name = n.attribute("name"), type = n.attribute("type"), tag = mcl::parse::to_lower(n.name);
            if (tag == "solver") {
                std::vector<mcl::Param> params;
                mcl::load_params(params, n);
                for (size_t i = 0; i < params.size(); ++i) {
                    if (params[i].tag == "iterations") system->settings.admm_iters = params[i].as_int();
                    else if (params[i].tag == "timestep") system->settings.timestep_s = params[i].as_double();
                    else if (params[i].tag == "realtime") settings.run_realtime = params[i].as_bool();
                    else if (params[i].tag == "verbose") system->settings.verbose = params[i].as_int();
                }
            } else if (tag == "force") {
                if (name.size() == 0 || type.size() == 0) throw std::runtime_error("\n**SimContext::load Error: Force \"" + tag + "\" need a name and type.");
                mcl::Component c2(tag, name, type);
                mcl::load_params(c2.params, n);
                force_param_map.insert(std::make_pair(name, c2));
            }
        }
        if (!scene->load(config_file)) throw std::runtime_error("\nExiting...");
    }

    // SimContext.cpp:103-169: forces that need the whole scene (gravity, XML anchors, wind over every
    // dynamic face), then System::initialize (device upload + factorisation).
    void initialize() {
        add_scene_forces();
        if (!system->initialize()) throw std::runtime_error("\nExiting...");
    }

    // first half of initialize(): host only, no device needed
    void add_scene_forces() {
        for (std::unordered_map<std::string, mcl::Component>::iterator it = force_param_map.begin(); it != force_param_map.end(); ++it) {
            const std::string type = mcl::parse::to_lower(it->second.type);
            if (type == "explicitforce") {
                std::shared_ptr<admm::ExplicitForce> ef(new admm::ExplicitForce());
                const trimesh::vec v = it->second.get("direction").as_vec3();
                for (int j = 0; j < 3; ++j) ef->direction[j] = v[j];
                system->explicit_forces.push_back(ef);
            } else if (type == "staticanchor") {
                system->forces.push_back(std::shared_ptr<admm::Force>(new admm::StaticAnchor(it->second.get("index").as_int())));
            } else if (type == "windforce" || type == "wind") {
                std::vector<int> faces = dynamic_faces();
                std::shared_ptr<admm::ExplicitForce> wf(new admm::WindForce(faces));
                const trimesh::vec v = it->second.get("direction").as_vec3();
                for (int j = 0; j < 3; ++j) wf->direction[j] = v[j];
                system->explicit_forces.push_back(wf);
            }
        }
    }

    // Node triples of every face of every object that has a <Force>, objects in object_params order,
    // offset by the vertex counts of the dynamic objects before it (SimContext.cpp:131-153; the same
    // loop windyflag.cpp:98-121 runs by hand).
    std::vector<int> dynamic_faces() const {
        std::vector<int> faces;
        int total = 0;
        for (std::unordered_map<std::string, std::vector<mcl::Param> >::const_iterator o = scene->object_params.begin(); o != scene->object_params.end(); ++o) {
            bool has_force = false;
            for (size_t p = 0; p < o->second.size(); ++p) if (o->second[p].tag == "force") has_force = true;
            if (!has_force) continue;
            std::shared_ptr<trimesh::TriMesh> mesh = scene->objects_map.at(o->first)->get_TriMesh();
            if (!mesh) throw std::runtime_error("\nSimContext::initialize Error: Problem with mesh creation.");
            for (size_t f = 0; f < mesh->faces.size(); ++f) for (int j = 0; j < 3; ++j) faces.push_back(mesh->faces[f][j] + total);
            total += (int)mesh->vertices.size();
        }
        return faces;
    }

    // SimContext.cpp:172-193: copy node positions back into the (float) mesh vertices
    bool update(mcl::SceneManager * = 0) {
        for (std::unordered_map<int, std::pair<int, int> >::const_iterator it = system_to_scene_map.begin(); it != system_to_scene_map.end(); ++it) {
            std::shared_ptr<trimesh::TriMesh> mesh = scene->objects[it->second.first]->get_TriMesh();
            if (!mesh) throw std::runtime_error("\nSimContext::update Error, something went wrong...");
            mesh->vertices[it->second.second] = trimesh::point((float)system->m_x[it->first * 3 + 0], (float)system->m_x[it->first * 3 + 1], (float)system->m_x[it->first * 3 + 2]);
        }
        for (size_t i = 0; i < scene->objects.size(); ++i) scene->objects[i]->update();
        return true;
    }

    // SimContext.cpp:196-210
    bool step(const mcl::SceneManager * = 0, float screen_dt = 0.f) {
        if (!settings.run_realtime) return system->step();
        double timeleft = screen_dt;
        while (timeleft > 0.0) {
            if (!system->step()) return false;
            timeleft -= system->settings.timestep_s;
        }
        return true;
    }

private:
    admm::ForceBuilder builder;
    std::unordered_map<int, std::pair<int, int> > system_to_scene_map;
    std::unordered_map<std::string, mcl::Component> force_param_map;
};
