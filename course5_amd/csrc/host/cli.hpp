// Boost-free parser for the reference's option table (main.cpp:17-69).
#pragma once

#include <iosfwd>

// Fills app::instance().config.  Returns true when rendering should proceed; false after
// --help or when -f/-d are missing (the caller then returns 0, as the reference does).
// Throws std::runtime_error on malformed options (boost::program_options throws there too).
bool program_options(int argc, char** argv, std::ostream& out);
void print_usage(std::ostream& out);
