// One reference training iteration (train.py:159-190) as a native launch sequence:
// mask -> x[mask]=0 -> forward -> MSE on masked nodes -> backward -> Adam.  See gatres_train_step_t.
#include "gatres_common.h"

extern "C" int gatres_train_step(const gatres_train_step_t* ts, void* stream) {
  if (!ts || !ts->graph || !ts->params || !ts->x || !ts->y || !ts->mask || !ts->out || !ts->g_out || !ts->loss ||
      !ts->saved || !ts->scratch)
    return GATRES_E_BADARG;
  const int N = ts->graph->num_nodes;
  int rc = 0;
  if ((ts->phases & GATRES_PHASE_MASK) && ts->node_ptr) {
    rc = gatres_mask_generate(ts->node_ptr, ts->num_graphs, ts->mask_rate, ts->seed, ts->step_counter, ts->mask,
                              stream);
    if (rc) return rc;
  }
  if (ts->phases & GATRES_PHASE_FORWARD) {
    rc = gatres_model_forward(&ts->model, ts->graph, ts->params, ts->x, ts->mask, ts->out, ts->saved, ts->scratch,
                              stream);
    if (rc) return rc;
    rc = gatres_masked_mse(ts->out, ts->y, ts->mask, ts->loss, ts->g_out, N, stream);
    if (rc) return rc;
  }
  if (ts->phases & GATRES_PHASE_BACKWARD) {
    if (!ts->grads) return GATRES_E_BADARG;
    rc = gatres_model_backward(&ts->model, ts->graph, ts->params, ts->x, ts->mask, ts->g_out, ts->saved, ts->scratch,
                               ts->grads, nullptr, stream);
    if (rc) return rc;
  }
  if (ts->phases & GATRES_PHASE_ADAM) {
    if (!ts->grads || !ts->exp_avg || !ts->exp_avg_sq || !ts->step_counter) return GATRES_E_BADARG;
    rc = gatres_adam_step(ts->params, ts->grads, ts->exp_avg, ts->exp_avg_sq, ts->step_counter,
                          gatres_param_count(ts->model.num_blocks, ts->model.nc), ts->lr, ts->beta1, ts->beta2, ts->eps,
                          ts->weight_decay, ts->grad_scale, stream);
    if (rc) return rc;
  }
  return 0;
}
