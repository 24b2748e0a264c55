// pg_gp.hip -- intentionally empty: gp::ols lives next to the streaming pass it shares (pg_sweep.hip).
#include "pg_common.h"
