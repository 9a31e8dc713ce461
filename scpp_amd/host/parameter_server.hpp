// Host mirror of the reference ParameterServer (scpp_core/utils/include/parameterServer.hpp:17-127): typed
// scalar / vector loading from Boost-INFO files (`key value`, `key { (i) value ... scaling s }`, `;` comments).
// Same member names and the same error texts; no Boost, no Eigen.
#pragma once
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace scpp
{

class ParameterServer
{
    struct Node
    {
        bool has_value = false;
        std::string value;
        std::map<std::string, std::shared_ptr<Node>> children;
    };
    Node root;

    struct Tok
    {
        std::string s;
        int line;
    };

public:
    explicit ParameterServer(const std::string &filename)
    {
        std::ifstream f(filename);
        if (!f)
            throw std::runtime_error("Could not open parameter file " + filename);
        std::vector<Tok> toks;
        std::string line;
        int ln = 0;
        while (std::getline(f, line))
        {
            const size_t sc = line.find(';');
            if (sc != std::string::npos)
                line.resize(sc);
            std::string cur;
            auto flush = [&]() {
                if (!cur.empty())
                    toks.push_back({cur, ln});
                cur.clear();
            };
            for (char ch : line)
            {
                if (ch == '{' || ch == '}')
                {
                    flush();
                    toks.push_back({std::string(1, ch), ln});
                }
                else if (ch == ' ' || ch == '\t' || ch == '\r')
                    flush();
                else
                    cur.push_back(ch);
            }
            flush();
            ln++;
        }
        std::vector<Node *> stack{&root};
        size_t i = 0;
        while (i < toks.size())
        {
            const Tok &t = toks[i];
            if (t.s == "}")
            {
                if (stack.size() < 2)
                    throw std::runtime_error("INFO parse error: unmatched } in " + filename);
                stack.pop_back();
                i++;
                continue;
            }
            if (t.s == "{")
                throw std::runtime_error("INFO parse error: unexpected { in " + filename);
            auto node = std::make_shared<Node>();
            stack.back()->children[t.s] = node;
            i++;
            if (i < toks.size() && toks[i].line == t.line && toks[i].s != "{" && toks[i].s != "}")
            {
                node->has_value = true;
                node->value = toks[i].s;
                i++;
            }
            if (i < toks.size() && toks[i].s == "{")
            {
                stack.push_back(node.get());
                i++;
            }
        }
    }

    template <typename T>
    void loadScalar(const std::string &name, T &scalar) const
    {
        auto it = root.children.find(name);
        if (it == root.children.end() || !it->second->has_value)
            throw std::runtime_error("WARNING: Failed to load scalar type: " + name + "!");
        parse(it->second->value, scalar, name);
    }

    // vector of `rows` entries `(i) value`, optional `scaling s`
    void loadMatrix(const std::string &name, double *out, int rows) const
    {
        auto it = root.children.find(name);
        if (it == root.children.end())
            throw std::runtime_error("Failed to load matrix type: " + name + "!");
        const auto &ch = it->second->children;
        double scaling = 1.;
        int entries = int(ch.size());
        auto sc = ch.find("scaling");
        if (sc != ch.end())
        {
            scaling = std::stod(sc->second->value);
            entries--;
        }
        if (entries < rows)
            throw std::runtime_error("Missing entries in matrix type: " + name + "!");
        if (entries > rows)
            throw std::runtime_error("Redundant entries in matrix type: " + name + "!");
        for (int i = 0; i < rows; i++)
        {
            auto e = ch.find("(" + std::to_string(i) + ")");
            if (e == ch.end())
                throw std::runtime_error("Failed to load matrix type: " + name + "!");
            out[i] = std::stod(e->second->value) * scaling;
        }
    }

private:
    static void parse(const std::string &v, bool &out, const std::string &name)
    {
        if (v == "true" || v == "1")
            out = true;
        else if (v == "false" || v == "0")
            out = false;
        else
            throw std::runtime_error("WARNING: Failed to load scalar type: " + name + "!");
    }
    template <typename T>
    static void parse(const std::string &v, T &out, const std::string &name)
    {
        try
        {
            out = T(std::stod(v));
        }
        catch (const std::exception &)
        {
            throw std::runtime_error("WARNING: Failed to load scalar type: " + name + "!");
        }
    }
};

} // namespace scpp
