// step_head_kernels.hip -- the fused VecTask step WITH the policy's output heads + sampling in its prologue (mms_bind_policy_head):
// the <MMS_TASK_TEN_ANT, 768, 16, 10, DR = false, HEAD = true> instantiation of step_kernels.hip's kernel template, in a translation
// unit of its own so that its code cannot perturb the register allocation of the other instantiations (see the note in
// step_kernels.hip at MMS_STEP_HEAD_TU).  Reference call sites it replaces: ActorCritic.act's last layers + `distribution.sample()`
// (algorithms/rl/ppo/module.py:80-100) followed by BaseTask.step (tasks/agent_base/base_task.py:129-149).
#define MMS_STEP_HEAD_TU 1
#include "step_kernels.hip"
