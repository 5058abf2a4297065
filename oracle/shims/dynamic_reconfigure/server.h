// Stand-in for <dynamic_reconfigure/server.h> — TEST INFRASTRUCTURE ONLY (oracle build).
#ifndef ORACLE_SHIM_DYNAMIC_RECONFIGURE_SERVER_H
#define ORACLE_SHIM_DYNAMIC_RECONFIGURE_SERVER_H
namespace dynamic_reconfigure
{
template <typename T>
class Server
{
};
}  // namespace dynamic_reconfigure
#endif
