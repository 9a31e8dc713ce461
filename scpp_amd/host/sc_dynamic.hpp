// Library entry point of the reference (scpp/include/sc_dynamic.hpp:6, scpp/src/sc_dynamic.cpp:3-15): one SC solve of the
// model's current configuration; plus the batched form the device engine exists for.
#pragma once
#include <memory>
#include <vector>

#include "sc_algorithm.hpp"

// B instances that differ in their initial state
inline std::vector<trajectory_data_t> sc_dynamic(std::shared_ptr<Model> model, const std::vector<Model::state_vector_t> &x_inits)
{
    scpp::SCAlgorithm solver(model, int(x_inits.size()));
    solver.initialize();
    scpp::batch_result_t r;
    solver.solveBatch(x_inits, r);
    return r.td;
}

// the reference's single-instance form: the model's configured initial state
inline trajectory_data_t sc_dynamic(std::shared_ptr<Model> model)
{
    return sc_dynamic(model, std::vector<Model::state_vector_t>{model->p.x_init}).front();
}
