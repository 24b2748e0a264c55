// pg_kinship_w8.hip -- the kinship kernel of pg_kinship.hip built for 8-wave workgroups (512 threads, 12 tile slots per wave),
// used for up to 64 pools, where 16 waves find no tiles to own.  Same source, same launcher logic; only the launcher is exported
// (pg_launch_kinship_w8), pg_launch_kinship hands over to it.
#define KIN_WAVES_DEF 8
#define KIN_LAUNCH_NAME pg_launch_kinship_w8
#define KIN_NO_EXPORTS
#include "pg_kinship.hip"
