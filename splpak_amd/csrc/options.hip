// Switches of the library: the table of known names, the process defaults, the per-thread current snapshot (options.hpp).
#include "options.hpp"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <mutex>

extern char **environ;

namespace splpak {

namespace {
struct Known { const char *name; int documented; };
const Known KNOWN[] = {
    {"SPLPAK_BIN_ATOMIC", 0},
    {"SPLPAK_DEBUG", 1},
    {"SPLPAK_DEBUG_NO_PEER", 0},
    {"SPLPAK_DEBUG_SUMS", 0},
    {"SPLPAK_DEBUG_TWOEND", 0},
    {"SPLPAK_DIST_CHUNK", 0},
    {"SPLPAK_EVAL_NO_PERSISTENT", 0},
    {"SPLPAK_EVAL_RUNS_MAXBINS", 0},
    {"SPLPAK_EVAL_SORT", 0},
    {"SPLPAK_GATHER_LOOKUP", 0},
    {"SPLPAK_GRAM_SCRATCH_MB", 1},
    {"SPLPAK_GRAM_VALU", 0},
    {"SPLPAK_MPLAN_BAND", 0},
    {"SPLPAK_MPLAN_RCCL", 1},
    {"SPLPAK_NARROW_BW", 0},
    {"SPLPAK_ND", 1},
    {"SPLPAK_ND_CHAIN_LA", 0},
    {"SPLPAK_ND_CHUNK", 0},
    {"SPLPAK_ND_CLEAR_WGS", 0},
    {"SPLPAK_ND_CUT", 1},
    {"SPLPAK_ND_DEBUG_STAGES", 0},
    {"SPLPAK_ND_DIST", 0},
    {"SPLPAK_ND_DUMMY_STREAMS", 0},
    {"SPLPAK_ND_FULL_DIAG", 0},
    {"SPLPAK_ND_HALVES", 0},
    {"SPLPAK_ND_JOIN_SQUARE", 0},
    {"SPLPAK_ND_KB", 1},
    {"SPLPAK_ND_NO_EARLY_CLEAR", 0},
    {"SPLPAK_ND_NO_FUSE", 0},
    {"SPLPAK_ND_NO_OUTER", 0},
    {"SPLPAK_ND_NO_ROOT_LOOKAHEAD", 0},
    {"SPLPAK_ND_PINNED_SPLIT", 0},
    {"SPLPAK_ND_PIN_FIRST", 0},
    {"SPLPAK_ND_PIN_ROUNDS", 0},
    {"SPLPAK_ND_POTRF_WAVES", 0},
    {"SPLPAK_ND_PREP_EARLY", 0},
    {"SPLPAK_ND_RES_CUS", 1},
    {"SPLPAK_ND_ROOT_LA", 0},
    {"SPLPAK_ND_SMALL_GRID", 0},
    {"SPLPAK_ND_SMALL_QUEUE", 0},
    {"SPLPAK_ND_SPLIT", 1},
    {"SPLPAK_ND_SQUARE", 0},
    {"SPLPAK_ND_STAGED_INIT", 0},
    {"SPLPAK_ND_WG4", 0},
    {"SPLPAK_ND_XCD", 0},
    {"SPLPAK_NO_CONSTRAINT_TABLE", 0},
    {"SPLPAK_NO_LOOKAHEAD", 0},
    {"SPLPAK_NO_NARROW", 0},
    {"SPLPAK_NO_PANEL_CU", 0},
    {"SPLPAK_NO_PLAN_CACHE", 1},
    {"SPLPAK_NO_REORDER", 1},
    {"SPLPAK_NO_STOPEV", 0},
    {"SPLPAK_NO_TWOEND", 0},
    {"SPLPAK_PCG_ALWAYS", 0},
    {"SPLPAK_PCG_ASSEMBLE", 0},
    {"SPLPAK_PCG_BLOCKS_F64", 0},
    {"SPLPAK_PCG_BLOCKS_UNPACKED", 0},
    {"SPLPAK_PCG_EAGER", 0},
    {"SPLPAK_PCG_MAXIT", 1},
    {"SPLPAK_PCG_NO_BLOCKS", 0},
    {"SPLPAK_PCG_NO_PAIRS", 0},
    {"SPLPAK_PCG_PAIRS_VALU", 0},
    {"SPLPAK_PCG_TOL1", 1},
    {"SPLPAK_PCG_TRI_PAIRS", 0},
    {"SPLPAK_PCG_TOL2", 1},
    {"SPLPAK_PIN_BW", 0},
    {"SPLPAK_PR_C0", 0},
    {"SPLPAK_PR_DEAL4", 0},
    {"SPLPAK_PR_NODEAL", 0},
    {"SPLPAK_RCCL_JOB", 0},
    {"SPLPAK_RCCL_LIB", 1},
    {"SPLPAK_RCCL_ONE_RANK_CALLS", 0},
    {"SPLPAK_RESIDUAL_CELLS", 0},
    {"SPLPAK_RESIDUAL_STAGED", 0},
    {"SPLPAK_ROWS_ONE_STREAM", 0},
    {"SPLPAK_ROWS_TILES", 0},
    {"SPLPAK_SOLVER", 1},
    {"SPLPAK_TOPA64", 0},
    {"SPLPAK_VIRTUAL_GPUS", 0},
};
std::mutex g_mu;
std::map<std::string, std::string> g_defaults;      // set through the API; "" with g_unset = hidden
std::map<std::string, bool> g_hidden;
thread_local const Options *t_current = nullptr;
thread_local Options t_fallback;
thread_local unsigned long long t_fallback_gen = ~0ull;
unsigned long long g_gen = 0;                        // bumped by every options_set_default
}  // namespace

bool option_canonical(const char *name, std::string &canon)
{
    if (!name) return false;
    std::string s(name);
    for (char &c : s) c = (char)std::toupper((unsigned char)c);
    if (s.rfind("SPLPAK_", 0) != 0) s = "SPLPAK_" + s;
    for (const Known &k : KNOWN)
        if (s == k.name) { canon = s; return true; }
    return false;
}

int option_documented(const std::string &canon)
{
    for (const Known &k : KNOWN)
        if (canon == k.name) return k.documented;
    return 0;
}

Options options_snapshot()
{
    Options o;
    for (char **e = environ; e && *e; ++e) {
        if (std::strncmp(*e, "SPLPAK_", 7) != 0) continue;
        const char *eq = std::strchr(*e, '=');
        if (!eq) continue;
        o.kv[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    for (const auto &h : g_hidden) o.kv.erase(h.first);
    for (const auto &d : g_defaults) o.kv[d.first] = d.second;
    return o;
}

int options_set_default(const char *name, const char *value)
{
    std::string canon;
    if (!option_canonical(name, canon)) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_gen;
    if (value) { g_defaults[canon] = value; g_hidden.erase(canon); }
    else { g_defaults.erase(canon); g_hidden.erase(canon); }
    return 0;
}

const char *opt_get(const char *canon)
{
    if (t_current) return t_current->get(canon);
    // outside of any plan: the process snapshot, refreshed when a default has changed (the environment itself is read per call,
    // as before -- this branch is not on the fit path)
    unsigned long long gen;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        gen = g_gen;
        auto d = g_defaults.find(canon);
        if (d != g_defaults.end()) {
            if (t_fallback_gen != gen) { t_fallback.kv.clear(); t_fallback_gen = gen; }
            t_fallback.kv[canon] = d->second;
            return t_fallback.kv[canon].c_str();
        }
    }
    return std::getenv(canon);
}

OptionsScope::OptionsScope(const Options *o) : prev(t_current) { t_current = o; }
OptionsScope::~OptionsScope() { t_current = prev; }

}  // namespace splpak
