// Switches of the library (round 6).  A plan takes a SNAPSHOT of them when it is created -- the process defaults set through
// splpak_set_default_option over the SPLPAK_* variables of the environment -- and everything a fit does reads that snapshot:
// no getenv on the fit path, a Fortran / C caller has an API for them (include/splpak_hip.h), and the one-shot entry's cached
// plan is keyed by the snapshot as a whole.  Inside a plan's calls the snapshot is the calling thread's current options
// (OptionsScope); outside of any plan (the evaluation entry points, the debug entries) opt_get reads the process snapshot.
#pragma once
#include <map>
#include <string>

namespace splpak {

struct Options {
    std::map<std::string, std::string> kv;          // canonical name ("SPLPAK_ND_KB") -> value
    const char *get(const char *name) const
    {
        auto it = kv.find(name);
        return it == kv.end() ? nullptr : it->second.c_str();
    }
    bool operator==(const Options &o) const { return kv == o.kv; }
};

// canonical name of a switch given as "nd_kb", "ND_KB" or "SPLPAK_ND_KB"; false = unknown
bool option_canonical(const char *name, std::string &canon);
// what kind of switch it is: 1 = one of the documented options, 0 = an internal A/B switch of the tests
int option_documented(const std::string &canon);
// process defaults over the environment
Options options_snapshot();
int options_set_default(const char *name, const char *value);      // value NULL: back to the environment's
// the calling thread's current options, else the process snapshot
const char *opt_get(const char *canon);
struct OptionsScope {
    const Options *prev;
    explicit OptionsScope(const Options *o);
    ~OptionsScope();
};

}  // namespace splpak
